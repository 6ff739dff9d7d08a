#!/usr/bin/env python3
"""Throughput of the any-label-count kernels (crf_general.hip) on a C2-shaped
batch (1 000 contigs, ~2e5 genes, A = 35 000) for L = 3, 8, 32, and of the same kernels forced
onto the 2-label model next to the specialised ones.  Run on the GPU box; prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from gecco_amd import _native as nat, synth  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    out = {}
    rng = np.random.default_rng(synth.SEED)
    A = 35000
    lengths = synth.contig_lengths(rng, 1000)
    cptr, gptr, attr = synth.synth_contigs(rng, lengths, A)
    n = int(cptr[-1])
    dev = torch.device("cuda:0")
    d_gp = torch.from_numpy(gptr).to(dev)
    d_at = torch.from_numpy(attr).to(dev)
    only = [int(v) for v in sys.argv[1:]]  # label counts to run (all, plus the long-contig cases, when none is given)
    # the same shape five times as large (5 000 contigs, ~1.1 M genes) for the windowed kernels: the C2-sized batch is
    # 940 workgroups of the matrix-core kernel -- 1.2 residency rounds, i.e. the second round runs a fifth full
    big = None
    if not only or os.environ.get("GECCO_BENCH_GENERAL_BIG", "1") == "1":
        rng_b = np.random.default_rng(synth.SEED + 1)
        bc, bg, ba = synth.synth_contigs(rng_b, synth.contig_lengths(rng_b, 5000), A)
        big = (bc, torch.from_numpy(bg).to(dev), torch.from_numpy(ba).to(dev), int(bc[-1]))
    for L in only or (2, 3, 4, 6, 8, 16, 32):
        if L == 2:
            w, trans = synth.synth_model(A, rng)
            os.environ["GECCO_CRF_FORCE_GENERAL"] = "1"
        else:
            os.environ.pop("GECCO_CRF_FORCE_GENERAL", None)
            w = np.clip(rng.laplace(0.0, 1.7, size=(A, L)), -6.3, 12.7)
            trans = rng.normal(0, 1.5, size=(L, L))
        model = nat.Model.from_tables(w, trans)
        plan = nat.Plan(model, cptr, 20, 1, True, device=0)
        p = torch.zeros(n, dtype=torch.float64, device=dev)
        y = torch.zeros(n, dtype=torch.int8, device=dev)
        marg = torch.zeros(n, L, dtype=torch.float64, device=dev)
        res = {"kernel": plan.kernel_name, "genes": n}
        for name, fn in (("windowed", lambda: plan.run_windowed(d_gp.data_ptr(), d_at.data_ptr(), p.data_ptr(), L - 1)),
                         ("viterbi", lambda: plan.run_viterbi(d_gp.data_ptr(), d_at.data_ptr(), y.data_ptr())),
                         ("marginals_full", lambda: plan.run_marginals_full(d_gp.data_ptr(), d_at.data_ptr(), marg.data_ptr()))):
            dt = timed(fn)
            res[name] = {"ms": dt * 1e3, "genes_per_s": n / dt}
        if big is not None and L > 2:
            bplan = nat.Plan(model, big[0], 20, 1, True, device=0)
            bp = torch.zeros(big[3], dtype=torch.float64, device=dev)
            dt = timed(lambda: bplan.run_windowed(big[1].data_ptr(), big[2].data_ptr(), bp.data_ptr(), L - 1))
            res["windowed_5000_contigs"] = {"genes": big[3], "ms": dt * 1e3, "genes_per_s": big[3] / dt}
        out[f"L={L}" + (" (2-label model forced onto the general kernels)" if L == 2 else "")] = res
    os.environ.pop("GECCO_CRF_FORCE_GENERAL", None)
    if only:
        print(json.dumps(out))
        return
    # long contigs: one 50 000-gene contig, contig-sequential kernels (one group of lanes walks it) against the
    # chunked ones (chunk matrices -> vectors over chunks -> replay inside chunks)
    lc, lg, la = synth.synth_contigs(rng, [50000], A)
    d_lg, d_la = torch.from_numpy(lg).to(dev), torch.from_numpy(la).to(dev)
    for L in (3, 8):
        w = np.clip(rng.laplace(0.0, 1.7, size=(A, L)), -6.3, 12.7)
        trans = rng.normal(0, 1.5, size=(L, L))
        model = nat.Model.from_tables(w, trans)
        res = {"genes": 50000}
        for mode in ("0", "1"):
            os.environ["GECCO_CRF_GENERAL_CHUNKED"] = mode
            plan = nat.Plan(model, lc, 20, 1, True, device=0)
            y = torch.zeros(50000, dtype=torch.int8, device=dev)
            marg = torch.zeros(50000, L, dtype=torch.float64, device=dev)
            tag = "chunked" if mode == "1" else "contig_sequential"
            for name, fn in (("viterbi", lambda: plan.run_viterbi(d_lg.data_ptr(), d_la.data_ptr(), y.data_ptr())),
                             ("marginals_full", lambda: plan.run_marginals_full(d_lg.data_ptr(), d_la.data_ptr(), marg.data_ptr()))):
                dt = timed(fn, reps=3)
                res[f"{name}_{tag}"] = {"ms": dt * 1e3, "genes_per_s": 50000 / dt}
        os.environ.pop("GECCO_CRF_GENERAL_CHUNKED", None)
        for name in ("viterbi", "marginals_full"):
            res[f"{name}_speedup"] = res[f"{name}_contig_sequential"]["ms"] / res[f"{name}_chunked"]["ms"]
        out[f"one 50000-gene contig, L={L}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
