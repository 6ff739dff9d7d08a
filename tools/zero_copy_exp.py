#!/usr/bin/env python3
"""Experiment: the windowed kernel reading the CSR from, and writing p to, PINNED HOST memory directly (no copies)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gecco_amd import _native as nat, synth  # noqa: E402
from oracle import crf_oracle as orc  # noqa: E402

wl = synth.workload("C3")
n = int(wl["contig_ptr"][-1])
model = nat.Model.from_tables(wl["w"], wl["trans"])
gp, at = nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
out = nat.pinned_empty(n, np.float64)
y = nat.pinned_empty(n, np.int8)
t0 = time.perf_counter()
plan = nat.Plan(model, wl["contig_ptr"], 20, 1, True, device=0)
print("plan create ms", (time.perf_counter() - t0) * 1e3)
for mode in ("windowed", "decode"):
    ts = []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "windowed":
            plan.run_windowed(gp.ctypes.data, at.ctypes.data, out.ctypes.data, 1, 0)
        else:
            plan.run_decode(gp.ctypes.data, at.ctypes.data, out.ctypes.data, y.ctypes.data, 1, 0, 0)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, "zero-copy ms:", " ".join(f"{t:.3f}" for t in ts), f"-> {n / min(ts) / 1e6:.2f} G genes/s")
exp = orc.windowed_marginals_mt(wl["w"], wl["trans"], wl["contig_ptr"], wl["gene_ptr"], wl["attr_id"], 20, 1, 1, True, threads=32)
print("max |dp|", float(np.abs(out - exp).max()))
ses = nat.Session(model, [0])
cp = nat.pinned_copy(wl["contig_ptr"])
ts = []
for _ in range(8):
    t0 = time.perf_counter()
    ses.windowed_marginals(cp, gp, at, 20, out=out)
    ts.append((time.perf_counter() - t0) * 1e3)
print("session (copies) ms:", " ".join(f"{t:.3f}" for t in ts))
