#!/usr/bin/env python3
"""Whole-contig marginals (row F) and Viterbi with path scores (row V, matrix form) on C3 and C5:
per-call time of the resident API.  Run on the GPU box; prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from gecco_amd import _native as nat, synth  # noqa: E402


def main():
    out = {}
    dev = torch.device("cuda:0")
    for name in ("C3", "C5"):
        wl = synth.workload(name)
        n, nc = int(wl["contig_ptr"][-1]), len(wl["contig_ptr"]) - 1
        model = nat.Model.from_tables(wl["w"], wl["trans"])
        plan = nat.Plan(model, wl["contig_ptr"], 20, 1, True, device=0)
        gp = torch.from_numpy(wl["gene_ptr"]).to(dev)
        at = torch.from_numpy(wl["attr_id"]).to(dev)
        marg = torch.zeros(n, 2, dtype=torch.float64, device=dev)
        ln = torch.zeros(nc, dtype=torch.float64, device=dev)
        y = torch.zeros(n, dtype=torch.int8, device=dev)
        sc = torch.zeros(nc, dtype=torch.float64, device=dev)
        res = {"genes": n}
        for key, fn in (("marginals_full", lambda: plan.run_marginals_full(gp.data_ptr(), at.data_ptr(), marg.data_ptr(), ln.data_ptr())),
                        ("marginals_full_without_log_z", lambda: plan.run_marginals_full(gp.data_ptr(), at.data_ptr(), marg.data_ptr())),
                        ("viterbi_with_scores", lambda: plan.run_viterbi(gp.data_ptr(), at.data_ptr(), y.data_ptr(), sc.data_ptr())),
                        ("viterbi_labels_only", lambda: plan.run_viterbi(gp.data_ptr(), at.data_ptr(), y.data_ptr()))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            res[key] = {"ms": dt * 1e3, "genes_per_s": n / dt}
        out[name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
