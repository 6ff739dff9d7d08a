#!/usr/bin/env python3
"""Do CU-masked streams (hipExtStreamCreateWithCUMask) partition the chip for our kernels?  Times the C3 window kernel and the C5
decode step on streams that see a subset of the CUs.  usage: cu_mask_probe.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gecco_amd import _native as nat, synth  # noqa: E402

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    """bits: iterable of CU indices (0 .. 255) the stream may use"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return s.value


dev = torch.device("cuda", 0)
torch.cuda.init()
torch.zeros(1, device=dev)
wl = synth.workload("C3")
model = nat.Model.from_tables(wl["w"], wl["trans"])
d_gp, d_at = torch.from_numpy(wl["gene_ptr"]).to(dev), torch.from_numpy(wl["attr_id"]).to(dev)
n = int(wl["contig_ptr"][-1])
plan = nat.Plan(model, wl["contig_ptr"], 20, 1, True, device=0)
d_p = torch.zeros(n, dtype=torch.float64, device=dev)
patterns = {
    "all 256": range(256),
    "first 128 (bits 0..127)": range(128),
    "even bits": range(0, 256, 2),
    "bits 0..15 of every 32": [b for b in range(256) if (b & 31) < 16],
    "first 64": range(64),
    "bits 0..7 of every 32 (64 CUs)": [b for b in range(256) if (b & 31) < 8],
}
for name, bits in patterns.items():
    s = masked_stream(bits)
    plan.time_windowed(d_gp.data_ptr(), d_at.data_ptr(), d_p.data_ptr(), 1, s, warmup=50, iters=200)
    ms = plan.time_windowed(d_gp.data_ptr(), d_at.data_ptr(), d_p.data_ptr(), 1, s, warmup=5, iters=300)
    print(f"C3 window kernel on a stream masked to {name}: {ms * 1e3:.1f} us")
