#!/usr/bin/env python3
"""One-shot (host buffers in / out) calls of the batch driver on C3, for a rocprofv3 kernel + memory-copy
trace: where does a call spend its time?  usage: trace_oneshot.py [chunk_genes] [reps] [pinned|pageable]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gecco_amd import _native as nat, synth  # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pinned = (sys.argv[3] if len(sys.argv) > 3 else "pinned") == "pinned"
wl = synth.workload("C3")
n = int(wl["contig_ptr"][-1])
model = nat.Model.from_tables(wl["w"], wl["trans"])
ses = nat.Session(model, [0])
ses.set_chunk_genes(chunk)
conv = nat.pinned_copy if pinned else np.array
cp, gp, at = conv(wl["contig_ptr"]), conv(wl["gene_ptr"]), conv(wl["attr_id"])
out = nat.pinned_empty(n, np.float64) if pinned else np.empty(n)
for _ in range(3):
    ses.windowed_marginals(cp, gp, at, 20, out=out)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    ses.windowed_marginals(cp, gp, at, 20, out=out)
    ts.append((time.perf_counter() - t0) * 1e3)
    time.sleep(0.002)
print("ms per call:", " ".join(f"{t:.3f}" for t in ts), "stats", ses.stats())
