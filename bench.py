#!/usr/bin/env python3
"""Benchmark of the GECCO CRF hot path on MI355X (contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one resident batch: the windowed forward-backward
marginals of every gene (the reference's `ClusterCRF.predict_probabilities` arithmetic,
gecco/crf/__init__.py:244-258) followed by whole-contig Viterbi decoding of the same batch
(one `gecco_crf_plan_run_decode`).  Workload at N=1 = BASELINE.json configs[2] ("C3": 10k-contig
synthetic metagenome, ~2M genes, 35k-attribute synthetic model, W=20).

N>1 (one process per GPU, no collective on the data path):
  * `value` is WEAK scaling -- every rank owns its own C3-sized batch of contigs;
  * `strong_scaling` in the same JSON line is BASELINE.json configs[3] ("C4"): the ONE C3 batch
    greedy-partitioned by gene count over the N devices (`gecco_amd.sharding.partition_contigs`),
    every rank scoring its own shard.
Inputs are resident in HBM before the timed region.  One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
SIMDS = 256 * 4         # 256 CUs x 4 SIMDs
SCLK_HZ = 2.4e9         # peak engine clock; an fp64 / DPP / 32-bit VALU instruction occupies a SIMD for 4 cycles per wave
VALU_SUSTAINED_CYCLES = 4.7  # ... nominally; measured on MI355X under sustained fp64 load: 4.6-4.8 (tools/ubench/valu_rates.hip)
W, STEP, LABEL = 20, 1, 1


def _alg_bytes(n_genes, nnz, n_contigs):
    # algorithmic bytes of one windowed launch: CSR row pointers + attribute ids in,
    # one fp64 probability per gene out, + the contig table; weight table excluded.
    return 4 * (n_genes + 1) + 4 * nnz + 8 * n_genes + 4 * (n_contigs + 1)


def _ratio_form_fallback_fraction(wl, plan_tiles, tile_out):
    """Fraction of the windowed kernel's workgroups that leave the 3+3-op ratio form of the DP because a slot in
    their reach leans further towards the label than exp(600/W) (crf_kernels.hip): computed on the host from the
    weights, for the report only (C3 has no padded contig: slot space = gene space)."""
    d_attr = wl["w"][:, LABEL] - wl["w"][:, 1 - LABEL]
    gp = wl["gene_ptr"].astype(np.int64)
    n = len(gp) - 1
    csum = np.concatenate([[0.0], np.cumsum(d_attr[wl["attr_id"]])])
    d = csum[gp[1:]] - csum[gp[:-1]]
    big = d > 600.0 / W
    if not big.any():
        return 0.0
    pre = np.concatenate([[0], np.cumsum(big)])
    t = np.arange(plan_tiles, dtype=np.int64)
    lo = np.clip(t * tile_out - (W - 1), 0, n)
    hi = np.clip((t + 1) * tile_out + (W - 1), 0, n)
    return float(((pre[hi] - pre[lo]) > 0).mean())


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C5", "Cinf"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-past-l3", action="store_true", help="skip the HBM-resident (2e8-gene) roofline point")
    ap.add_argument("--kernel-iters", type=int, default=200)
    ap.add_argument("--windowed-only", action="store_true",
                    help="steps are plain windowed launches (for PMC passes: one kind of dispatch)")
    ap.add_argument("--preroll-ms", type=float, default=30.0,
                    help="untimed device pre-roll before the warmup steps: the GPU needs ~10 ms of sustained work to reach steady clocks")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # GECCO_BENCH_ONE_DEVICE=1 (dry runs of the N > 1 code path on a one-GPU box): every rank on device 0, gloo
    one_device = os.environ.get("GECCO_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    red_dev = dev
    if world > 1:
        import torch.distributed as dist

        if one_device:
            dist.init_process_group(backend="gloo")
            red_dev = torch.device("cpu")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from gecco_amd import _native as nat
    from gecco_amd import sharding, synth

    def barrier():
        if dist is not None:
            dist.barrier()

    class Resident:
        """One batch resident on this rank's device: plan + CSR + outputs."""

        def __init__(self, model, cptr, gptr, attr):
            self.n_genes, self.nnz, self.n_contigs = int(cptr[-1]), int(gptr[-1]), len(cptr) - 1
            self.plan = nat.Plan(model, cptr, W, STEP, True, device=local_rank)
            self.d_gp = torch.from_numpy(np.ascontiguousarray(gptr)).to(dev)
            self.d_at = torch.from_numpy(np.ascontiguousarray(attr) if len(attr) else np.zeros(1, np.int32)).to(dev)
            self.d_p = torch.zeros(max(self.n_genes, 1), dtype=torch.float64, device=dev)
            self.d_y = torch.zeros(max(self.n_genes, 1), dtype=torch.int8, device=dev)
            # a stream of its own, not the legacy default one: the decode step is replayed as a HIP graph, and streams
            # are captured into graphs everywhere but there
            if os.environ.get("GECCO_BENCH_OWN_STREAM", "1") == "1":
                self.torch_stream = torch.cuda.Stream(dev)
                self.stream = self.torch_stream.cuda_stream
            else:
                self.stream = torch.cuda.current_stream(dev).cuda_stream

        def step(self):
            if args.windowed_only:
                self.plan.run_windowed(self.d_gp.data_ptr(), self.d_at.data_ptr(), self.d_p.data_ptr(), LABEL, self.stream)
            else:  # one pass over the CSR: state scores are accumulated once for both outputs
                self.plan.run_decode(self.d_gp.data_ptr(), self.d_at.data_ptr(), self.d_p.data_ptr(), self.d_y.data_ptr(), LABEL, 0,
                                     self.stream)

        def timed(self, steps, warmup, preroll_ms):
            """(seconds for `steps` steps, max over ranks), after `warmup` untimed ones"""
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < preroll_ms:
                for _ in range(20):
                    self.step()
                torch.cuda.synchronize(dev)
            for _ in range(warmup):
                self.step()
            barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize(dev)
            barrier()
            elapsed = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            return elapsed

    def all_sum(v):
        if dist is None:
            return int(v)
        g = torch.tensor([int(v)], dtype=torch.int64, device=red_dev)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return int(g.item())

    # ---- workload: every rank generates its own batch (same model, different contigs) for the weak-scaling value
    wl = synth.workload(args.workload, seed=synth.SEED)
    base = dict(wl)  # rank 0's batch = the batch BASELINE.json names; C4 partitions THIS one
    if rank > 0:
        rng = np.random.default_rng(synth.SEED + rank)
        lengths = np.diff(wl["contig_ptr"]).astype(np.int64)
        rng.shuffle(lengths)
        hot = np.argsort(wl["w"][:, 1] - wl["w"][:, 0])[-200:]
        cptr, gptr, attr = synth.synth_contigs(rng, lengths, wl["A"], planted=0.01, hot_attrs=hot)
        wl.update(contig_ptr=cptr, gene_ptr=gptr, attr_id=attr)
    model = nat.Model.from_tables(wl["w"], wl["trans"])
    res = Resident(model, wl["contig_ptr"], wl["gene_ptr"], wl["attr_id"])
    n_genes, nnz = res.n_genes, res.nnz

    elapsed = res.timed(args.steps, args.warmup, args.preroll_ms)
    total_genes = all_sum(n_genes)

    # ---- dominant kernel: average launch duration by HIP events on the launch stream
    kern_ms = res.plan.time_windowed(res.d_gp.data_ptr(), res.d_at.data_ptr(), res.d_p.data_ptr(), LABEL, res.stream, warmup=3,
                                     iters=args.kernel_iters)
    alg_bytes = _alg_bytes(n_genes, nnz, res.n_contigs)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    # PMC figures of this kernel on this workload, from the committed profile of the same command
    # (tools/profile.sh -> tools/pmc_to_json.py); null when there is none
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path)).get(args.workload, {}) or {}
        except Exception:
            pmc = {}
    traffic = pmc.get("hbm_bytes_per_launch")
    valu_insts = pmc.get("SQ_INSTS_VALU")  # wave-level VALU instructions of one launch
    valu_frac = (valu_insts * 4.0 / (kern_ms * 1e-3 * SIMDS * SCLK_HZ)) if valu_insts else None

    out = {
        "metric": "genes/sec CRF decode (windowed fwd-bwd marginals" + (")" if args.windowed_only else " + Viterbi)"),
        "value": total_genes * args.steps / elapsed,
        "unit": "genes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {res.n_contigs} contigs, {n_genes} genes, {nnz} domain hits per GPU; "
                        f"A=35000 synthetic 2-label model, window 20 step 1, pad.  Deviation from SURVEY.md 8d: weights "
                        f"~ Laplace(-0.4, 1.7) (8d: location 0) and the Zipf head (ids < A/50) forced negative, so that "
                        f"most genes lean to label '0' as under the embedded model (mean w1-w0 = -0.76)",
            "genes_per_gpu": n_genes,
            "viterbi_in_step": not args.windowed_only,
            "device_preroll_ms": args.preroll_ms,
            "sharding": "independent contig batches per rank, no collective",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": res.plan.kernel_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": pmc.get("source"),
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms": kern_ms,
            # the limiter in practice: fp64 VALU issue.  wave-level VALU instructions of one launch (PMC
            # SQ_INSTS_VALU, same profile) x 4 cycles each / (kernel time x 1024 SIMDs x 2.4 GHz)
            "valu_frac": valu_frac,
            "valu_insts_per_launch": valu_insts,
            # the same against the issue rate this GPU SUSTAINS on fp64 (tools/ubench/valu_rates.hip, four waves per
            # SIMD of independent chains: 4.6-4.8 "2.4 GHz cycles" per wave instruction for v_add/v_mul/v_fma_f64 and the
            # DPP moves, i.e. ~2.05 GHz effective under this load) instead of the nominal 4 cycles at 2.4 GHz
            "valu_frac_at_sustained_rate": (valu_frac * VALU_SUSTAINED_CYCLES / 4.0) if valu_frac else None,
            "ratio_form_fallback_frac": _ratio_form_fallback_fraction(wl, res.plan.num_tiles, 2 * (256 - (W - 1)))
            if args.workload != "Cinf" else None,
        },
    }

    # ---- C4: the ONE base batch partitioned over the ranks (strong scaling, BASELINE.json configs[3])
    if world > 1:
        lengths = np.diff(base["contig_ptr"]).astype(np.int64)
        mine = sharding.partition_contigs(lengths, world)[rank]
        cptr, gptr, attr, _ = sharding.extract_shard(base["contig_ptr"], base["gene_ptr"], base["attr_id"], mine)
        shard = Resident(model, cptr, gptr, attr)
        el = shard.timed(args.steps, args.warmup, 0.0)
        tot = all_sum(shard.n_genes)
        out["strong_scaling"] = {
            "config": f"C4: the {args.workload} batch ({tot} genes) greedy-partitioned by gene count over {world} devices "
                      f"(gecco_amd.sharding.partition_contigs), no collective",
            "value": tot * args.steps / el, "unit": "genes/s", "ms_per_step": el / args.steps * 1e3,
            "genes_on_rank0": shard.n_genes, "scaling": "strong",
        }

    # ---- the HBM-resident point: C3 sits in the 256 MiB Infinity Cache across repeated steps; 2e8 genes do not
    if rank == 0 and world == 1 and not args.no_past_l3 and args.workload == "C3":
        try:
            big = synth.workload("Cinf", seed=synth.SEED)
            rb = Resident(model, big["contig_ptr"], big["gene_ptr"], big["attr_id"])
            ms = rb.plan.time_windowed(rb.d_gp.data_ptr(), rb.d_at.data_ptr(), rb.d_p.data_ptr(), LABEL, rb.stream, warmup=2, iters=10)
            ab = _alg_bytes(rb.n_genes, rb.nnz, rb.n_contigs)
            out["roofline_past_l3"] = {
                "workload": f"Cinf: {rb.n_genes} genes (100 x C3), {ab / 1e9:.2f} GB algorithmic per launch: past the 256 MiB Infinity Cache",
                "kernel_ms": ms, "achieved": ab / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "genes_per_s": rb.n_genes / (ms * 1e-3),
            }
            del rb, big
            torch.cuda.empty_cache()
        except Exception as err:  # a smaller device, a busy box: the headline numbers do not depend on this point
            out["roofline_past_l3"] = {"error": str(err)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (C restatement of the CRFsuite tagger driven window by
        # window like the reference), 1 thread, on a bounded sample of the same workload.
        from oracle import crf_oracle as orc

        nc = min(len(wl["contig_ptr"]) - 1, 10000)
        cp = wl["contig_ptr"][: nc + 1]
        ng = int(cp[-1])
        t0 = time.perf_counter()
        p_ref = orc.windowed_marginals(wl["w"], wl["trans"], cp, wl["gene_ptr"][: ng + 1], wl["attr_id"], W, STEP, LABEL, True)
        dt_win = time.perf_counter() - t0
        t0 = time.perf_counter()
        y_ref, _ = orc.viterbi(wl["w"], wl["trans"], cp, wl["gene_ptr"][: ng + 1], wl["attr_id"])
        dt_vit = time.perf_counter() - t0
        dt = dt_win + (0.0 if args.windowed_only else dt_vit)
        res.step()
        torch.cuda.synchronize(dev)
        got = res.d_p[:ng].cpu().numpy()
        out["cpu_baseline"] = {
            "value": ng / dt,
            "unit": "genes/s",
            "cores": 1,
            "kind": "port",
            "sample": f"first {nc} contigs ({ng} genes) of the same workload: windowed marginals {dt_win:.2f} s"
                      + ("" if args.windowed_only else f" + Viterbi {dt_vit:.2f} s") + ", C oracle driven window by window like the reference",
        }
        # SURVEY.md 8d also asks for the same restatement on all host cores (contigs over threads)
        ncpu = os.cpu_count() or 1
        best = None
        for _ in range(2):  # first threaded pass wakes the cores up
            t0 = time.perf_counter()
            orc.windowed_marginals_mt(wl["w"], wl["trans"], cp, wl["gene_ptr"][: ng + 1], wl["attr_id"], W, STEP, LABEL, True, threads=ncpu)
            if not args.windowed_only:
                orc.viterbi_mt(wl["w"], wl["trans"], cp, wl["gene_ptr"][: ng + 1], wl["attr_id"], threads=ncpu)
            d = time.perf_counter() - t0
            best = d if best is None else min(best, d)
        out["cpu_baseline_all_cores"] = {"value": ng / best, "unit": "genes/s", "cores": ncpu, "kind": "port",
                                         "sample": "same sample, contigs spread over all host threads, best of 2"}
        out["parity"] = {
            "max_abs_dp_vs_oracle": float(np.abs(got - p_ref).max()),
            "cluster_call_mismatches": int(((got > 0.8) != (p_ref > 0.8)).sum()),
            "genes_checked": ng,
        }
        if not args.windowed_only:
            out["parity"]["viterbi_label_mismatches"] = int((res.d_y[:ng].cpu().numpy() != y_ref.astype(np.int8)).sum())
        # the TRUE reference, only if the box happens to have it (a site install; never shipped from
        # this repository): sklearn_crfsuite's tagger in the reference's own per-window Python loop
        # (gecco/crf/__init__.py:251-256) on the embedded model, 10^4 genes, one core
        ref = _true_reference_baseline()
        if ref is not None:
            out["cpu_baseline_port"] = out["cpu_baseline"]
            out["cpu_baseline"] = ref
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _true_reference_baseline():
    try:
        import sklearn_crfsuite  # noqa: F401
    except Exception:
        return None
    try:
        import pickle

        from gecco_amd import synth

        golden = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(golden, "model.pkl"), "rb") as fh:
            crf = pickle.load(fh)  # needs gecco + sklearn_crfsuite importable: the reference's own class
        tagger = crf.model
        rng = np.random.default_rng(synth.SEED)
        attrs = list(tagger.attributes_)
        cptr, gptr, attr = synth.synth_contigs(rng, [200] * 50, len(attrs))
        feats = [{attrs[a]: True for a in attr[gptr[g]:gptr[g + 1]]} for g in range(int(cptr[-1]))]
        t0 = time.perf_counter()
        n_win = 0
        for c in range(len(cptr) - 1):
            seq = feats[cptr[c]:cptr[c + 1]]
            probs = np.zeros(len(seq))
            for s in range(len(seq) - W + 1):  # the reference's loop, gecco/crf/__init__.py:251-256
                marg = tagger.predict_marginals_single(seq[s:s + W])
                probs[s:s + W] = np.maximum(probs[s:s + W], [m["1"] for m in marg])
                n_win += 1
        dt = time.perf_counter() - t0
        n = int(cptr[-1])
        return {"value": n / dt, "unit": "genes/s", "cores": 1, "kind": "reference",
                "sample": f"sklearn_crfsuite on the embedded model, 50 contigs x 200 genes ({n_win} windows) in {dt:.1f} s, "
                          f"the reference's per-window Python loop (marginals only)"}
    except Exception:
        return None


if __name__ == "__main__":
    main()
