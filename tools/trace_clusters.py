#!/usr/bin/env python3
"""Cluster calls (marginals + refiner on the device, rows back) of the batch driver on C3, pinned buffers:
per-kernel durations (run under rocprofv3 --kernel-trace --stats) and GECCO_CRF_TRACE=1 host timings.
usage: trace_clusters.py [chunk_genes] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gecco_amd import _native as nat, synth  # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = synth.workload("C3")
n = int(wl["contig_ptr"][-1])
model = nat.Model.from_tables(wl["w"], wl["trans"])
ses = nat.Session(model, [0])
ses.set_chunk_genes(chunk)
cp, gp, at = (nat.pinned_copy(wl[k]) for k in ("contig_ptr", "gene_ptr", "attr_id"))
ann = nat.pinned_copy((np.diff(wl["gene_ptr"]) > 0).astype(np.uint8))
for _ in range(3):
    ses.clusters(cp, gp, at, ann, 20, want_p=False, want_seg_p=True)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    seg = ses.clusters(cp, gp, at, ann, 20, want_p=False, want_seg_p=True)[0]
    ts.append((time.perf_counter() - t0) * 1e3)
    time.sleep(0.002)
print("ms per call:", " ".join(f"{t:.3f}" for t in ts), "rows", len(seg), "stats", ses.stats())
