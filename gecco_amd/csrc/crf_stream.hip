// Streaming form of the windowed-marginals kernel (row W of SURVEY.md §8a, gecco/crf/__init__.py:244-258) for the
// headline shape: two labels, W = 20, no rescaling.  Same arithmetic as `crf_windowed_l2` (crf_kernels.hip) -- one lane = one window start, ratio-form
// recurrences, Z-constant candidates, DPP diagonal maximum -- but organised as a software pipeline:
//
//   * a workgroup owns PH consecutive phases of NT window starts and walks them in order.  The running maxima that
//     leave the last lane of a phase are handed to the first lanes of the NEXT phase (through LDS, like the hand-over
//     between the waves of one phase), so only the first W-1 starts of the whole workgroup are lead-in work: a
//     workgroup produces NT*PH - (W-1) outputs for NT*PH window starts (the tiled kernel: NT - (W-1) per NT).
//   * stage 1 of phase c+1 runs UNDER the DP of phase c-1 without holding a single VGPR: row pointers, attribute ids
//     and the gathered weight pairs travel HBM/L2 -> LDS with gfx950's LDS-direct buffer loads
//     (`buffer_load_dword / dwordx4 ... lds`: lane l lands at M0 + l * size; out-of-range offsets land as zeros), which
//     are fire-and-forget until `s_waitcnt vmcnt(0)`.  What the tiled kernel pays per workgroup -- three dependent memory
//     round trips before the first DP step -- is paid once per workgroup here and one (already landed) wait per phase.
//   * barriers are LDS-only (`s_waitcnt lgkmcnt(0); s_barrier`): `__syncthreads()` would drain vmcnt, i.e. wait for
//     the very loads that are meant to stay in flight under the DP.
//   * the slot constants r = mu01 exp(s[label] - s[other]) of the whole workgroup range stay in LDS (8 B per slot);
//     the parking area for the weight pairs is reused phase by phase.  20.4 KB of LDS: eight workgroups per CU.
//   * register discipline: at eight waves per SIMD a wave has 64 VGPRs and 80 SGPRs, and what the compiler cannot keep
//     in SGPRs it parks in VGPR lanes (v_writelane / v_readlane: VALU slots this kernel does not have to spare).  So
//     the first and the last iteration of the phase loop are peeled (everything that depends on the phase KIND is a
//     compile-time constant), and the kernel arguments only one section needs are re-read from the kernel-argument
//     segment where they are used (scalar loads, scalar-cache hits) instead of living in SGPRs across the loop.
// Sums are in CSR order (bit-exact with sequential addition), as in the tiled kernel.
#include "crf_device.hpp"

namespace gecco {
namespace {

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));
// pointers that come out of an asm statement are generic: cast to the global address space for global_load / global_store
// (a flat access also occupies lgkmcnt and checks the LDS aperture)
#define GLOBAL_PTR(T, p) (reinterpret_cast<__attribute__((address_space(1))) T *>(reinterpret_cast<uintptr_t>(p)))

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Kernel arguments that only one section of the phase loop needs are RE-READ from the kernel-argument segment where
// they are used (scalar loads, scalar-cache hits) instead of living in SGPRs across the whole loop.  The compiler does
// not know these loads: they are issued by one asm statement and waited for by another one that takes the registers
// as read-write operands, so that no use can move above the wait.
struct StageArgs {  // stage 1
    double c6, c5, c4, c3, c2;   // 1/6!, 1/5!, 1/4!, 1/3!, 1/2!  (WinArgs::expc[7..11])
    double ratio_dmax;
    double *dstate_out;
    const uint32_t *start_bits;
    const double *rtab;
};
__device__ __forceinline__ void load_stage_args(StageArgs &t) {
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("s_load_dwordx2 %0, %9, %10\n\ts_load_dwordx2 %1, %9, %10+8\n\ts_load_dwordx2 %2, %9, %10+16\n\t"
                 "s_load_dwordx2 %3, %9, %10+24\n\ts_load_dwordx2 %4, %9, %10+32\n\ts_load_dwordx2 %5, %9, %11\n\t"
                 "s_load_dwordx2 %6, %9, %12\n\ts_load_dwordx2 %7, %9, %13\n\ts_load_dwordx2 %8, %9, %14"
                 : "=&s"(t.c6), "=&s"(t.c5), "=&s"(t.c4), "=&s"(t.c3), "=&s"(t.c2), "=&s"(t.ratio_dmax), "=&s"(t.dstate_out),
                   "=&s"(t.start_bits), "=&s"(t.rtab)
                 : "s"(ka), "n"(offsetof(WinArgs, expc) + 56), "n"(offsetof(WinArgs, ratio_dmax)), "n"(offsetof(WinArgs, dstate_out)),
                   "n"(offsetof(WinArgs, start_bits)), "n"(offsetof(WinArgs, rtab))
                 : "memory");
}
__device__ __forceinline__ void wait_stage_args(StageArgs &t) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(t.c6), "+s"(t.c5), "+s"(t.c4), "+s"(t.c3), "+s"(t.c2), "+s"(t.ratio_dmax), "+s"(t.dstate_out),
                   "+s"(t.start_bits), "+s"(t.rtab)::"memory");
}
struct SlowArgs {  // irregular workgroups only: the contig table of the plan, the workgroup's descriptor, the row pointers
    const int32_t *c_slot, *c_gene, *c_n, *gene_ptr;
    const int4 *tile_desc;
};
__device__ __forceinline__ SlowArgs load_slow_args() {
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    SlowArgs t;
    asm volatile("s_load_dwordx2 %0, %5, %6\n\ts_load_dwordx2 %1, %5, %7\n\ts_load_dwordx2 %2, %5, %8\n\ts_load_dwordx2 %3, %5, %9\n\t"
                 "s_load_dwordx2 %4, %5, %10\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t.c_slot), "=&s"(t.c_gene), "=&s"(t.c_n), "=&s"(t.tile_desc), "=&s"(t.gene_ptr)
                 : "s"(ka), "n"(offsetof(WinArgs, c_slot)), "n"(offsetof(WinArgs, c_gene)), "n"(offsetof(WinArgs, c_n)),
                   "n"(offsetof(WinArgs, tile_desc)), "n"(offsetof(WinArgs, gene_ptr))
                 : "memory");
    return t;
}
__device__ __forceinline__ double *load_p_out() {
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    double *p;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(p) : "s"(ka), "n"(offsetof(WinArgs, p_out)) : "memory");
    return p;
}
// mu01 exp(x) = 2^e (mu01 2^(j/32)) exp(r),  n = rint(x 32 / ln2) = 32 e + j,  r = x - n ln2 / 32 in two pieces
// (|r| <= ln2 / 64), degree-6 Taylor polynomial (truncation 3e-18 relative), table entry from `rtab` (L1-resident,
// requested before the polynomial), v_ldexp_f64 for 2^e (overflows to +inf / flushes to 0 by itself).  17 VALU
// instructions and 10 SGPRs of coefficients, against 28 and 24 for the degree-13 polynomial of exp_signed.
__device__ __forceinline__ double mu_exp(double x, const StageArgs &sa) {
    const double n = rint(x * 46.166241308446828384);   // 32 / ln 2
    const int ni = int(n);
    const double t = GLOBAL_PTR(const double, sa.rtab)[ni & 31];
    double r = fma(-n, 0.021660849392498290, x);         // ln2 / 32, high part
    r = fma(-n, 7.247021293269686e-19, r);               // low part
    double p = fma(sa.c6, r, sa.c5);
    p = fma(p, r, sa.c4);
    p = fma(p, r, sa.c3);
    p = fma(p, r, sa.c2);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p * t, ni >> 5);
}

__device__ __forceinline__ double wave_shr1_zero(double v) {  // lane l <- lane l-1, lane 0 <- +0.0
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double max_nocanon(double a, double b) {  // finite, non-negative operands
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {  // contiguous range of workgroups per XCD (shared halo in one L2)
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

template <int W, int NT, int PH, bool LABEL1>
struct Stream {
    static constexpr int NW = NT / 64;
    static constexpr int NR = NT * PH + W - 1;      // slots the workgroup needs constants for
    static constexpr int OUTW = NT * PH - (W - 1);  // output slots of the workgroup
    static constexpr int APL = 2, PCAP = APL * NT;  // weight pairs parked per round
    struct Smem {
        double *R;        // [NR] slot constants of the workgroup's range: |R| = r, sign bit set = a window may start here
        f64x2 *PARK;      // [PCAP] weight pairs of the phase in flight
        int32_t *IDS;     // [PCAP] its attribute ids
        int32_t *GP;      // [NT+1] its row pointers
        double *CARRY;    // [(NW+1)(W-1)] rows 0 .. NW-2: wave -> next wave; rows NW-1, NW: last wave -> next phase (by parity)
        uint32_t *AB;     // [PH+2] gene_ptr at the first gene of every stage phase (and behind the last one)
    };
    // what survives from phase to phase (all wave-uniform except tid)
    struct State {
        int tid, wave;
        int B, S;            // slot of r-index 0; slots of the batch
        int shift;           // regular workgroup: gene = slot + shift throughout its reach
        int irregular, wg;   // a padded or skipped contig in reach: slots are looked up one by one (slow, rare)
        int force_renorm;    // GECCO_CRF_RATIO=0: every wave takes the max-normalised form
        double rho, mu01, kappa_over_mu01, inv_kappa;
        __amdgpu_buffer_rsrc_t ra, rw, rg;  // attribute ids, weight pairs, row pointers (whole arrays: bounds-checked)
        // The probability a DP phase produces is STORED at the top of the next iteration, behind that iteration's
        // `s_waitcnt vmcnt(0)`: issued at the end of its own iteration, the store would be the youngest vector memory
        // operation in flight when the next iteration waits for its (long landed) stage-1 loads, and every phase would
        // pay a store round trip.
        double out_p;
        int out_g;  // its gene; -1: nothing pending
    };
    static __device__ __forceinline__ void flush_output(State &x) {
        if (x.out_g >= 0) GLOBAL_PTR(double, load_p_out())[x.out_g] = x.out_p;
        x.out_g = -1;
    }

    // ---- asynchronous stage-1 traffic: global -> LDS, no VGPR held
    static __device__ __forceinline__ void issue_ids(const State &x, const Smem &m, uint32_t a_base) {
#pragma unroll
        for (int a = 0; a < APL; ++a)  // the first PCAP attribute ids from position a_base
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x.ra, (lds_ptr)(m.IDS + a * NT + x.wave * 64), 4,
                                                     int((a_base + uint32_t(a * NT + x.tid)) << 2), 0, 0, 0);
    }
    // row pointers of the NT slots from slot q0 on (+ one behind): GP[j] = gene_ptr[gene of slot q0 + j]; slots outside
    // the batch take the first / last row pointer (an empty run)
    static __device__ __forceinline__ void issue_gp(const State &x, const Smem &m, int q0) {
        const int g = min(max(q0 + x.tid, 0), x.S) + x.shift;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x.rg, (lds_ptr)(m.GP + x.wave * 64), 4, g << 2, 0, 0, 0);
        if (x.tid == 0) {
            const int ge = min(max(q0 + NT, 0), x.S) + x.shift;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x.rg, (lds_ptr)(m.GP + NT), 4, ge << 2, 0, 0, 0);
        }
    }
    static __device__ __forceinline__ void issue_gathers(const State &x, const Smem &m, uint32_t n_attr) {
#pragma unroll
        for (int a = 0; a < APL; ++a) {  // weight pairs of the ids in IDS -> PARK (ids outside the dictionary: zeros)
            const uint32_t k = uint32_t(a * NT + x.tid);
            if (k < n_attr) {
                const uint32_t wo = min(uint32_t(m.IDS[k]), 0x0FFFFFFFu) << 4;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x.rw, (lds_ptr)(m.PARK + a * NT + x.wave * 64), 16, int(wo), 0, 0, 0);
            }
        }
    }

    // One iteration of the phase loop: stage 1 of phase c, then the DP of phase c-1 (none in the FIRST iteration; the
    // LAST iteration's stage covers only the W-1 slots behind the last window start and requests nothing further).
    template <bool FIRST, bool LAST>
    static __device__ __forceinline__ void iteration(State &x, const Smem &m, const int c) {
        // the thread and wave indices are opaque from here on: whatever is derived from them (lane predicates, LDS
        // addresses for M0) is recomputed in the iteration -- one SALU / VALU instruction each -- instead of being hoisted
        // out of the loop and parked in VGPR lanes across it
        asm volatile("" : "+v"(x.tid), "+s"(x.wave));
        const int tid = x.tid, wave = x.wave, lane = tid & 63;
        constexpr int ns = LAST ? W - 1 : NT;
        // ================= stage 1 of phase c: slot constants -> R =================
        wait_vm0();  // gathers(c) -> PARK, row pointers(c) -> GP, ids(c+1) -> IDS have landed
        lds_barrier();
        if (!FIRST) flush_output(x);  // (the previous iteration's DP phase)
        {
            const int i = NT * c + tid;  // r-index of the lane's slot
            const int q = x.B + i;       // the slot
            const bool in_range = (LAST ? tid < ns : true) && q >= 0 && q < x.S;
            StageArgs sa;
            load_stage_args(sa);  // (land under the sums)
            double s0 = 0.0, s1 = 0.0;
            int g = -1;  // the slot's gene (none: padding, or outside the batch)
            if (!x.irregular) {
                if (in_range) g = q + x.shift;
                const uint32_t lo = uint32_t(m.GP[tid]), hi = uint32_t(m.GP[tid + 1]);
                const uint32_t a_cur = m.AB[c];
                const uint32_t n_attr = __builtin_amdgcn_readfirstlane(m.AB[c + 1] - a_cur);
#pragma unroll 1
                for (uint32_t base = 0;; base += PCAP) {
                    const uint32_t c0 = a_cur + base, c1 = c0 + PCAP;
                    // the run [lo, hi) cut to this round, as addresses in the parking area
                    const f64x2 *k = m.PARK + (max(lo, c0) - c0);
                    const f64x2 *const e = m.PARK + (min(hi, c1) - c0);  // (k >= e when the run lies outside this round)
                    for (; k < e && hi > c0; ++k) {
                        const f64x2 v = *k;
                        s0 += v.x;
                        s1 += v.y;
                    }
                    if (base + PCAP >= n_attr) break;
                    // rare: more than PCAP attributes in one phase -- further rounds, synchronously
                    lds_barrier();
#pragma unroll
                    for (int a = 0; a < APL; ++a) {
                        const uint32_t kk = base + PCAP + uint32_t(a * NT + tid);
                        if (kk < n_attr) {
                            const int id = __builtin_amdgcn_raw_buffer_load_b32(x.ra, int((a_cur + kk) << 2), 0, 0);
                            const uint32_t wo = min(uint32_t(id), 0x0FFFFFFFu) << 4;
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(x.rw, (lds_ptr)(m.PARK + a * NT + wave * 64), 16, int(wo), 0, 0, 0);
                        }
                    }
                    wait_vm0();
                    lds_barrier();
                }
            } else {
                // Irregular workgroup (a padded or skipped contig in reach: rare): every slot looks its gene up in the
                // contig table of the workgroup's reach (crf/__init__.py:216-227: delta // 2 empty items in front) and
                // gathers its own attributes, synchronously.  The genes of the phase's slots are parked for the DP's
                // stores (GP / IDS by phase parity: nothing is in flight into them in this kind of workgroup).
                const SlowArgs t = load_slow_args();
                if (in_range) {
                    const int4 td = t.tile_desc[x.wg];
                    int lo = 0, hi = td.z - td.y;  // largest k with c_slot[k] <= q among the contigs in reach
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (t.c_slot[td.y + mid] <= q) lo = mid; else hi = mid - 1;
                    }
                    const int k = td.y + lo;
                    const int cs = t.c_slot[k], pos = q - cs, np = t.c_slot[k + 1] - cs, n = t.c_n[k];
                    const int gl = pos - ((np - n) >> 1);
                    if (gl >= 0 && gl < n) g = t.c_gene[k] + gl;
                }
                (c & 1 ? m.IDS : m.GP)[tid] = g;
                if (g >= 0) {
                    const uint32_t lo = uint32_t(t.gene_ptr[g]), hi = uint32_t(t.gene_ptr[g + 1]);
                    for (uint32_t k = lo; k < hi; ++k) {
                        const int id = __builtin_amdgcn_raw_buffer_load_b32(x.ra, int(k << 2), 0, 0);
                        const i32x4 w = __builtin_amdgcn_raw_buffer_load_b128(x.rw, int(min(uint32_t(id), 0x0FFFFFFFu) << 4), 0, 0);
                        s0 += __hiloint2double(w.y, w.x);
                        s1 += __hiloint2double(w.w, w.z);
                    }
                }
            }
            const double d = s1 - s0;
            wait_stage_args(sa);
            // a window may start at this slot: bit q of the plan's bit array (zero words in front of slot 0 and behind
            // the last slot, so any slot of the reach indexes it), carried in the sign bit of the slot constant
            const uint32_t sbit = (GLOBAL_PTR(const uint32_t, sa.start_bits)[q >> 5] >> (q & 31)) & 1u;
            // decode = windowed marginals + Viterbi of the same batch: the score differences of the genes this workgroup
            // owns are handed to the whole-contig kernel instead of being gathered again
            if (g >= 0 && sa.dstate_out && i >= W - 1 && i < W - 1 + OUTW) GLOBAL_PTR(double, sa.dstate_out)[g] = LABEL1 ? d : -d;
            const double r = mu_exp(d, sa);
            if (!LAST || tid < ns) m.R[i] = __hiloint2double(__double2hiint(r) | int(sbit << 31), __double2loint(r));
            x.force_renorm = sa.ratio_dmax < 0.0;  // (GECCO_CRF_RATIO=0)
        }
        lds_barrier();  // R published; PARK, GP and (per wave) IDS are free again
        const int cd = c - 1;  // the DP phase whose constants are now complete: r-indices [NT cd, NT cd + NT + W-1)
        double Rb = 0.0;       // running best of the lane's output slot
        const double rho = x.rho;
        // where lane 63 parks the running maximum that leaves the wave at step k: row `wave`, or for the last wave the
        // row of this phase's parity (read by the first wave of the next phase).  One address register for all steps.
        // (an LDS address as a 32-bit register: a generic pointer that went through an asm statement would make every one
        // of these stores a flat_store through the vector memory pipeline)
        typedef __attribute__((address_space(3))) double lds_double;
        uint32_t crow_a = uint32_t(reinterpret_cast<uintptr_t>((lds_ptr)(m.CARRY + (wave < NW - 1 ? wave : NW - 1 + (cd & 1)) * (W - 1))));
        asm volatile("" : "+v"(crow_a));
        lds_double *const crow = reinterpret_cast<lds_double *>(crow_a);
        // ================= stage 1 of phase c+1 leaves now and lands under the DP below =================
        if (!LAST && !x.irregular) {
            const uint32_t a_nxt = m.AB[c + 1], a_nx2 = m.AB[c + 2];
            issue_gathers(x, m, a_nx2 - a_nxt);  // (reads the ids of phase c+1)
            issue_ids(x, m, a_nx2);               // (phase c+2; past the last phase: never used)
            issue_gp(x, m, x.B + NT * (c + 1));
        }
        // ================= DP of phase c-1, ratio form =================
        if (!FIRST) {
            // every window in the ratio form first; a wave one of whose windows ends on Z >= 1e250 repeats the phase in the
            // max-normalised form, pairs (e0, f) derived from r on the fly (see crf_kernels.hip)
            const double *rrs = m.R + NT * cd + tid;
            double A1[W];
            const double r0 = rrs[0];
            double a0 = 1.0, a1 = fabs(r0) * x.kappa_over_mu01;
            A1[0] = a1;
#pragma unroll
            for (int k = 1; k < W; ++k) {
                const double r = fabs(rrs[k]);
                const double t = a0 + a1;
                a1 = fma(a1, rho, a0) * r;
                a0 = t;
                A1[k] = a1;
            }
            asm volatile("" ::: "memory");  // re-read the slot constants in the backward pass (VGPRs)
            double z = fma(a1, x.inv_kappa, a0);
            const bool renorm = __builtin_amdgcn_ballot_w64(!(z < 1.0e250) || x.force_renorm) != 0;
            auto pair_of = [&](double r, double &e0, double &f) {
                const bool pos = r > x.mu01;
                const double rc = fmin(r, 1.0e300);  // (r = inf for d > 709: e0 = 0 as exp(-d) would be)
                double q = __builtin_amdgcn_rcp(rc);
                q = fma(fma(-rc, q, 1.0), q, q);  // one Newton step: a division's worth of digits, a third of its registers
                e0 = pos ? x.mu01 * q : 1.0;
                f = pos ? x.mu01 : r;
            };
            if (renorm) {
                double e0, f;
                pair_of(fabs(r0), e0, f);
                a0 = e0;
                a1 = f * x.kappa_over_mu01;
                A1[0] = a1;
#pragma unroll  // (a rolled loop would index A1 dynamically and send the whole array -- the hot path's too -- to scratch)
                for (int k = 1; k < W; ++k) {
                    pair_of(fabs(rrs[k]), e0, f);
                    const double t = a0 + a1;
                    a1 = fma(a1, rho, a0) * f;
                    a0 = t * e0;
                    A1[k] = a1;
                }
                z = fma(a1, x.inv_kappa, a0);
            }
            double b0, b1;
            {
                double r = __builtin_amdgcn_rcp(z);
                r = fma(fma(-z, r, 1.0), r, r);
                b0 = r0 < 0.0 ? r : 0.0;  // (the sign bit: a window may start here)
                b1 = b0 * x.inv_kappa;
            }
            if (!renorm) {
#pragma unroll
                for (int k = W - 1; k >= 0; --k) {
                    const double cand = A1[k] * b1;
                    if (k < W - 1) {
                        if (lane == 63) crow[k] = Rb;
                        Rb = wave_shr1_zero(Rb);
                    }
                    Rb = max_nocanon(Rb, cand);
                    if (k > 0) {
                        const double u = fabs(rrs[k]) * b1;
                        b1 = fma(u, rho, b0);
                        b0 = b0 + u;
                    }
                }
            } else {
#pragma unroll
                for (int k = W - 1; k >= 0; --k) {
                    const double cand = A1[k] * b1;
                    if (k < W - 1) {
                        if (lane == 63) crow[k] = Rb;
                        Rb = wave_shr1_zero(Rb);
                    }
                    Rb = max_nocanon(Rb, cand);
                    if (k > 0) {
                        double e0, f;
                        pair_of(fabs(rrs[k]), e0, f);
                        const double ce = e0 * b0, u = f * b1;
                        b0 = ce + u;
                        b1 = fma(u, rho, ce);
                    }
                }
            }
            lds_barrier();
            // running maxima that left the previous wave -- or, for the first wave, the last wave of the previous phase
            const int prow = wave > 0 ? wave - 1 : NW - 1 + ((cd + 1) & 1);
            if (lane < W - 1 && (wave > 0 || cd > 0)) Rb = fmax(Rb, m.CARRY[prow * (W - 1) + lane]);
            Rb = fmin(Rb, 1.0);  // x_k and Z are rounded independently: x_k / Z may land one ulp above 1
            // the first W-1 slots of the workgroup's first phase lack the windows that start before it: the previous
            // workgroup writes them
            const int my_q = x.B + NT * cd + tid;
            int my_g = my_q >= 0 && my_q < x.S ? my_q + x.shift : -1;
            if (x.irregular) my_g = (cd & 1 ? m.IDS : m.GP)[tid];
#if defined(GECCO_STREAM_DEBUG) && GECCO_STREAM_DEBUG == 1  // experiment builds: the slot constants / the start flags instead
            Rb = fabs(m.R[NT * cd + tid]);
#elif defined(GECCO_STREAM_DEBUG) && GECCO_STREAM_DEBUG == 2
            Rb = m.R[NT * cd + tid] < 0.0 ? 1.0 : 0.0;
#endif
            x.out_p = Rb;
            x.out_g = (cd > 0 || tid >= W - 1) ? my_g : -1;
        }
    }
};

#ifndef GECCO_STREAM_OCC
#define GECCO_STREAM_OCC 8  // waves per SIMD the register allocation must allow (A/B builds: tools/build_variant.sh)
#endif
template <int W, int NT, int PH, bool LABEL1>
__global__ void __launch_bounds__(NT, GECCO_STREAM_OCC) crf_windowed_stream_l2(const WinArgs P) {
    using K = Stream<W, NT, PH, LABEL1>;
    __shared__ double R[K::NR];
    __shared__ __attribute__((aligned(16))) f64x2 PARK[K::PCAP];
    __shared__ int32_t IDS[K::PCAP];
    __shared__ int32_t GP[NT + 1];
    __shared__ double CARRY[(K::NW + 1) * (W - 1)];
    __shared__ uint32_t AB[PH + 2];
    const typename K::Smem m{R, PARK, IDS, GP, CARRY, AB};

    typename K::State x;
    x.tid = threadIdx.x;
    x.wave = __builtin_amdgcn_readfirstlane(x.tid >> 6);
    x.S = P.S;
    x.wg = xcd_remap(blockIdx.x, P.ntiles);
    x.B = x.wg * K::OUTW - (W - 1);
    {
        // (gene - slot shift, first contig, last contig, flags: 1 = regular); a batch without any padded or skipped contig
        // has slot space = gene space everywhere and no descriptor to wait for
        const int4 td = P.all_regular ? make_int4(0, 0, 0, 1) : P.tile_desc[x.wg];
        x.shift = td.x;
        x.irregular = (td.w & 1) ? 0 : 1;
        if (x.irregular) x.shift = 0;
    }
    const uint32_t nnz = uint32_t(P.gene_ptr[P.n_genes]);
    x.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(P.attr_id), 0, min(nnz, 0x3FFFFFFFu) << 2, 0x00020000);
    x.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(P.wtab2), 0, uint32_t(P.A) << 4, 0x00020000);
    x.rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(P.gene_ptr), 0, uint32_t(P.n_genes + 1) << 2, 0x00020000);
    x.rho = P.rho;
    x.mu01 = P.mu01;
    x.kappa_over_mu01 = P.kappa_over_mu01;
    x.inv_kappa = P.inv_kappa;
    // four separate register pairs, not sub-registers of the 8-dword tuple the argument load produced: the tuple is
    // spilled and reloaded as a whole (eight v_readlane per use of rho)
    asm volatile("" : "+s"(x.rho), "+s"(x.mu01), "+s"(x.kappa_over_mu01), "+s"(x.inv_kappa));
    x.force_renorm = 0;
    x.out_g = -1;
    x.out_p = 0.0;

    if (!x.irregular) {
        // gene_ptr at the first gene of stage phase c (c <= PH: slot B + NT c; c == PH + 1: behind the last slot of the reach)
        if (x.tid <= PH + 1) {
            const int q = x.B + NT * min(x.tid, PH) + (x.tid > PH ? W - 1 : 0);
            AB[x.tid] = uint32_t(P.gene_ptr[min(max(q, 0), x.S) + x.shift]);
        }
        const uint32_t a0 = uint32_t(P.gene_ptr[max(x.B, 0) + x.shift]);  // (phase 0; a scalar load)
        K::issue_ids(x, m, a0);
        K::issue_gp(x, m, x.B);
    }
    wait_vm0();
    lds_barrier();
    if (!x.irregular) {
        const uint32_t a0 = AB[0], a1 = AB[1];
        K::issue_gathers(x, m, a1 - a0);
        // (a wave only ever overwrites the part of IDS that its own lanes have just read: no barrier before the next ids)
        K::issue_ids(x, m, a1);
    }

    K::template iteration<true, false>(x, m, 0);
#pragma unroll 1
    for (int c = 1; c < PH; ++c) K::template iteration<false, false>(x, m, c);
    K::template iteration<false, true>(x, m, PH);
    K::flush_output(x);
}

}  // namespace

int windowed_stream_tile_out(int W, int phases) { return kWinThreads * phases - (W - 1); }

// Shapes the streaming kernel takes; everything else goes to the tiled kernel (crf_kernels.hip).  Byte offsets into
// the CSR arrays are 32-bit buffer offsets: batches of 2^30 genes / attribute ids and more stay with the tiled kernel
// (the plan checks the sizes).
bool windowed_stream_ok(const WinArgs &a) {
    return a.L == 2 && a.W == 20 && a.rescale_mask == 0 && !a.state_out && !a.generic && a.rtab;
}

template <int PH>
static void launch_stream_ph(const WinArgs &a, hipStream_t stream) {
    const dim3 grid(a.ntiles), block(kWinThreads);
    if (a.label)
        hipLaunchKernelGGL((crf_windowed_stream_l2<20, kWinThreads, PH, true>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((crf_windowed_stream_l2<20, kWinThreads, PH, false>), grid, block, 0, stream, a);
}

hipError_t launch_windowed_stream(const WinArgs &a, int phases, hipStream_t stream) {
    if (a.ntiles <= 0) return hipSuccess;
    if (!windowed_stream_ok(a)) return hipErrorNotSupported;
    switch (phases) {
    case 2: launch_stream_ph<2>(a, stream); break;
    case 3: launch_stream_ph<3>(a, stream); break;
    case 4: launch_stream_ph<4>(a, stream); break;
    default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

}  // namespace gecco
