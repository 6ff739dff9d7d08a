#!/usr/bin/env python3
"""What ONE step costs the host: wall time of the enqueue loop (no synchronisation inside) per call, for a trivial ctypes call, a
plain windowed launch and the pipelined decode launch, on a batch so small that the GPU is never the limit (4 streams)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gecco_amd import _native as nat, synth  # noqa: E402

dev = torch.device("cuda", 0)
genes = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(1)
w, trans = synth.synth_model(35000, rng) if hasattr(synth, "synth_model") else (None, None)
wl = synth.workload("C2")
model = nat.Model.from_tables(wl["w"], wl["trans"])
nc = int(np.searchsorted(wl["contig_ptr"], genes))
cptr = wl["contig_ptr"][: nc + 1]
n = int(cptr[-1])
gptr = wl["gene_ptr"][: n + 1]
attr = wl["attr_id"][: int(gptr[-1])]
d_gp, d_at = torch.from_numpy(np.ascontiguousarray(gptr)).to(dev), torch.from_numpy(np.ascontiguousarray(attr)).to(dev)
lanes = []
for k in range(4):
    s = torch.cuda.Stream(dev)
    lanes.append(dict(plan=nat.Plan(model, cptr, 20, 1, True, device=0), p=torch.zeros(n, dtype=torch.float64, device=dev),
                      y=torch.zeros(n, dtype=torch.int8, device=dev), ts=s, s=s.cuda_stream))
a_gp, a_at = d_gp.data_ptr(), d_at.data_ptr()
for ln in lanes:
    ln["ap"], ln["ay"] = ln["p"].data_ptr(), ln["y"].data_ptr()
    ln["plan"].run_decode_pipelined(a_gp, a_at, ln["ap"], None, 0, 1, ln["s"])
torch.cuda.synchronize()
K = 4000


def loop(fn, nl):
    for i in range(200):
        fn(lanes[i % nl])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        fn(lanes[i % nl])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6


print(f"batch: {n} genes, {lanes[0]['plan'].num_tiles} tiles")
print("trivial ctypes call (plan.num_tiles): issue %.2f us" % loop(lambda ln: ln["plan"].num_tiles, 1)[0])
for nl in (1, 2, 4):
    i, t = loop(lambda ln: ln["plan"].run_windowed(a_gp, a_at, ln["ap"], 1, ln["s"]), nl)
    print(f"run_windowed, {nl} streams: issue {i:.2f} us per call, with the final wait {t:.2f}")
    i, t = loop(lambda ln: ln["plan"].run_decode_pipelined(a_gp, a_at, ln["ap"], ln["plan"], ln["ay"], 1, ln["s"]), nl)
    print(f"run_decode_pipelined, {nl} streams: issue {i:.2f} us per call, with the final wait {t:.2f}")
    for ln in lanes:
        ln["call"] = ln["plan"].bind_decode_pipelined(a_gp, a_at, ln["ap"], ln["plan"], ln["ay"], 1, ln["s"])
    i, t = loop(lambda ln: ln["call"](), nl)
    print(f"the same through Plan.bind_decode_pipelined (arguments converted once), {nl} streams: issue {i:.2f} us per call, with the final wait {t:.2f}")
