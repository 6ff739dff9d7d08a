#!/usr/bin/env python3
"""Reference-bits mode against the fast kernels on C3 (and C1): the resident kernels (plan level; the mode comes from
GECCO_CRF_REFERENCE_BITS, so the script runs itself twice), the one-shot pinned call of a session in both modes, and the
levels of the drop-in class (tables, objects) in both modes.  Run on the GPU box; prints JSON lines."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from gecco_amd import _native as nat, synth  # noqa: E402


def resident():
    wl = synth.workload("C3")
    n = int(wl["contig_ptr"][-1])
    model = nat.Model.from_tables(wl["w"], wl["trans"])
    dev = torch.device("cuda", 0)
    plan = nat.Plan(model, wl["contig_ptr"], 20, 1, True, device=0)
    gp = torch.from_numpy(wl["gene_ptr"]).to(dev)
    at = torch.from_numpy(wl["attr_id"]).to(dev)
    p = torch.zeros(n, dtype=torch.float64, device=dev)
    ms = plan.time_windowed(gp.data_ptr(), at.data_ptr(), p.data_ptr(), 1, 0, warmup=3, iters=20)
    print(json.dumps({"resident_windowed_ms": ms, "genes": n, "genes_per_s": n / ms * 1e3, "kernel": plan.kernel_name,
                      "mode": "reference bits" if os.environ.get("GECCO_CRF_REFERENCE_BITS") == "1" else "fast kernels"}))


def timed(fn, reps=10):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def session_levels():
    wl = synth.workload("C3")
    n = int(wl["contig_ptr"][-1])
    model = nat.Model.from_tables(wl["w"], wl["trans"])
    ses = nat.Session(model, [0])
    cp, gpp, atp = nat.pinned_copy(wl["contig_ptr"]), nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
    outp = nat.pinned_empty(n, np.float64)
    res = {}
    for mode in (False, True, False, True):
        ses.set_reference_bits(mode)
        dt = timed(lambda: ses.windowed_marginals(cp, gpp, atp, 20, out=outp))
        res.setdefault("reference_bits" if mode else "fast", []).append(round(dt * 1e3, 4))
    print(json.dumps({"one_shot_pinned_ms": res, "genes": n}))
    from benchkit import latency

    blob_model = nat.Model.from_lcrf(latency.real_blob())
    ses1 = nat.Session(blob_model, [0])
    cptr, gptr, attr = latency.c1_batch(50, blob_model.num_attrs)
    res = {}
    for mode in (False, True, False, True):
        ses1.set_reference_bits(mode)
        dt = timed(lambda: ses1.windowed_marginals(cptr, gptr, attr, 20), reps=400)
        res.setdefault("reference_bits" if mode else "fast", []).append(round(dt * 1e6, 2))
    print(json.dumps({"c1_one_shot_us": res}))


def class_levels():
    from benchkit import levels

    golden = os.path.join(ROOT, "tests", "golden")
    for mode in ("0", "1"):
        os.environ["GECCO_AMD_REFERENCE_BITS"] = mode
        t = levels.tables_level(golden)
        o = levels.object_level(golden)
        print(json.dumps({"class_mode": "reference bits" if mode == "1" else "fast kernels",
                          "tables_genes_per_s": t.get("genes_per_s"), "tables_ms": t.get("ms"),
                          "objects_genes_per_s": o["genes_per_s"], "objects_breakdown_us_per_gene": o["breakdown_us_per_gene"]}))
    del os.environ["GECCO_AMD_REFERENCE_BITS"]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "resident":
        resident()
    else:
        for mode in ("0", "1"):
            env = dict(os.environ, GECCO_CRF_REFERENCE_BITS=mode)
            subprocess.run([sys.executable, os.path.abspath(__file__), "resident"], env=env, check=False)
        session_levels()
        class_levels()
