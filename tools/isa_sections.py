#!/usr/bin/env python3
"""Per-section instruction count of the window kernel (crf_windowed_l2<20, exact, 256 lanes, 2 tiles>): static ISA counts of
every straight-line region, multiplied by how often a lane runs it on a batch with `d` domains per gene, summed into the
sections DESIGN.md 4.1 names, and held against the counter (SQ_INSTS_VALU of one C3 launch, profiles/pmc_traffic.json).
usage: tools/isa_sections.py [--asm file.s]   (compiles gecco_amd/csrc/crf_kernels.hip for gfx950 when no file is given)"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN5gecco12_GLOBAL__N_115crf_windowed_l2ILi20ELb1ELb0ELi256ELi2EEEvNS_7WinArgsE"


def kernel_body(path):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
    return lines[start + 1:end]


def classify(t):
    op = t.split()[0]
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    return "salu"


def regions(body):
    """straight-line regions between labels / branches / barriers: (first line, last line, counts, terminator)"""
    out, cur, first = [], {"valu": 0, "lds": 0, "vmem": 0, "smem": 0, "salu": 0}, 0
    for i, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith(";") or (t.startswith(".") and not re.match(r"^\.LBB", t)):
            continue
        if re.match(r"^\.LBB", t):
            out.append((first, i, cur, t))
            cur, first = {k: 0 for k in cur}, i
            continue
        cur[classify(t)] += 1
        if t.startswith(("s_barrier", "s_cbranch", "s_branch", "s_endpgm")):
            out.append((first, i, cur, t))
            cur, first = {k: 0 for k in cur}, i + 1
    return out


def main():
    asm = None
    if "--asm" in sys.argv:
        asm = sys.argv[sys.argv.index("--asm") + 1]
    tmp = None
    if asm is None:
        tmp = tempfile.TemporaryDirectory()
        asm = os.path.join(tmp.name, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I",
                               os.path.join(ROOT, "include"), os.path.join(ROOT, "gecco_amd", "csrc", "crf_kernels.hip"), "-o", asm],
                              stderr=subprocess.DEVNULL)
    body = kernel_body(asm)
    regs = regions(body)
    # landmarks: the phase loop starts behind the last barrier of stage 1 (the first region that ends on a barrier followed by an
    # unconditional branch into a loop header); the forward pass is the first > 50-VALU region of the loop, the renorm path
    # holds the v_rcp_f64 of the per-slot reciprocals, the backward pass is the > 100-VALU region behind it
    big = [(k, r) for k, r in enumerate(regs) if r[2]["valu"] >= 50]
    # the two largest-with-DPP regions: forward (no DPP) and backward (DPP moves)
    def has(r, pat):
        return any(pat in body[i] for i in range(r[0], r[1] + 1))
    fwd = next(k for k, r in big if not has(r, "_dpp") and not has(r, "v_rcp_f64") and has(r, "ds_read_b64"))
    bwd = next(k for k, r in big if has(r, "wave_shr") and not has(r, "v_rcp_f64"))
    stage1_end = max(k for k, r in enumerate(regs[:fwd]) if r[3].startswith("s_barrier"))
    # the irregular path of stage 1 (a padded / skipped contig in reach: every slot looks its gene up and gathers eight weight
    # pairs) is not what a metagenome batch runs: the regions from the first to the last one with eight or more vector loads
    heavy = [k for k, r in enumerate(regs[:stage1_end + 1]) if r[2]["vmem"] >= 8]
    irregular = set(range(heavy[0], heavy[-1] + 1)) if heavy else set()
    # ... and the contig-table search in front of them (the first barrier of the kernel belongs to it)
    first_barrier = min(k for k, r in enumerate(regs) if r[3].startswith("s_barrier"))
    lookup = set()
    for k in range(first_barrier):  # the uniform branch that skips the search: everything up to its target label
        t = regs[k][3]
        if t.startswith("s_cbranch_vcc") or t.startswith("s_cbranch_scc"):
            label = t.split()[-1] + ":"
            j = next((j for j in range(k + 1, len(regs)) if regs[j][3].startswith(label)), None)
            if j is not None and j > first_barrier:
                lookup = set(range(k + 1, j + 1))
                break
    stage1 = [r for k, r in enumerate(regs[:stage1_end + 1]) if k not in irregular and k not in lookup]
    idx1 = [k for k in range(stage1_end + 1) if k not in irregular and k not in lookup]
    k_round = next(k for k in idx1 if has(regs[k], "buffer_load_dword "))    # the attribute round begins
    k_sums = max(k for k in idx1 if has(regs[k], "ds_read_b128"))              # ... the last parked-pair loop
    sections = {"stage 1a: slot -> gene, window-start bits, row bounds, descriptors": [regs[k] for k in idx1 if k < k_round],
                "stage 1b: attribute-per-lane ids + weight gathers, parking, every slot's sum (1.41 domains a gene)": [regs[k] for k in idx1 if k_round <= k <= k_sums + 2],
                "stage 1c: slot constant r = mu01 exp(d) (table + degree-6 polynomial), sign packing, hand-over store": [regs[k] for k in idx1 if k > k_sums + 2],
                "phase prologue (output gene, first slot constant)": regs[stage1_end + 1:fwd],
                "forward pass (19 steps x 3)": [regs[fwd]],
                "Z reciprocal, fallback vote": [r for r in regs[fwd + 1:bwd] if not has(r, "v_rcp_f64") or r[2]["valu"] < 12][:2],
                "backward pass + diagonal maximum (20 steps x 7)": [regs[bwd]],
                "epilogue (carry, clamp, store)": regs[bwd + 1:bwd + 6]}
    W, NT, TT = 20, 256, 2
    out_per_wg = TT * (NT - (W - 1))
    res = {}
    for name, rs in sections.items():
        tot = {k: sum(r[2][k] for r in rs) for k in ("valu", "lds", "vmem", "salu")}
        res[name] = tot
    # trip counts per workgroup-lane: stage 1 once (its attribute loop: one round; per-attribute inner loops ~1.41 trips per slot
    # are inside the static count once each: reported as static), phases TT times
    per_gene = {}
    for name, tot in res.items():
        mult = 1 if name.startswith("stage 1") else TT
        per_gene[name] = {k: v * mult * NT / out_per_wg for k, v in tot.items()}
    total_valu = sum(v["valu"] for v in per_gene.values())
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("C3", {})
    except Exception:
        pass
    print(f"window kernel, static ISA regions x trip counts per OUTPUT gene ({out_per_wg} genes per {NT}-lane workgroup, {TT} phases):")
    print(f"{'section':106s} {'VALU':>7s} {'LDS':>6s} {'VMEM':>6s} {'SALU':>6s}")
    for name, v in per_gene.items():
        print(f"{name:106s} {v['valu']:7.1f} {v['lds']:6.1f} {v['vmem']:6.1f} {v['salu']:6.1f}")
    print(f"{'sum (loops of stage 1 counted once: one attribute round, one parked pair per slot)':106s} {total_valu:7.1f}")
    if pmc.get("SQ_INSTS_VALU"):
        genes = 1999989
        print(f"counter: SQ_INSTS_VALU = {pmc['SQ_INSTS_VALU']:.0f} wave instructions per C3 launch = {pmc['SQ_INSTS_VALU'] * 64 / genes:.1f} lane "
              f"instructions per gene ({pmc.get('source', '')})")
    if tmp:
        tmp.cleanup()


if __name__ == "__main__":
    main()
