#!/usr/bin/env python3
"""Experiment: the pipelined decode on ONE stream against two independent pipelined decode streams (two plans, batches
alternating between them), so that the tail of one launch overlaps the head of the next.  Prints us per batch."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gecco_amd import _native as nat, synth  # noqa: E402


def main():
    wl = synth.workload("C3")
    dev = torch.device("cuda:0")
    model = nat.Model.from_tables(wl["w"], wl["trans"])
    n = int(wl["contig_ptr"][-1])
    gp = torch.from_numpy(wl["gene_ptr"]).to(dev)
    at = torch.from_numpy(wl["attr_id"]).to(dev)
    for ns in (1, 2, 3):
        plans = [nat.Plan(model, wl["contig_ptr"], 20, 1, True, device=0) for _ in range(ns)]
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        ps = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(ns)]
        ys = [torch.zeros(n, dtype=torch.int8, device=dev) for _ in range(ns)]
        primed = [False] * ns

        def step(i):
            k = i % ns
            plans[k].run_decode_pipelined(gp.data_ptr(), at.data_ptr(), ps[k].data_ptr(), plans[k] if primed[k] else None,
                                          ys[k].data_ptr() if primed[k] else 0, 1, streams[k].cuda_stream)
            primed[k] = True

        def flush():
            for k in range(ns):
                if primed[k]:
                    plans[k].flush_decode_pipelined(ys[k].data_ptr(), streams[k].cuda_stream)
                    primed[k] = False

        for i in range(300):
            step(i)
        flush()
        torch.cuda.synchronize()
        K = 2000
        t0 = time.perf_counter()
        for i in range(K):
            step(i)
        flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{ns} stream(s): {dt / K * 1e6:.2f} us per batch, {n * K / dt / 1e9:.1f} G genes/s")


if __name__ == "__main__":
    main()
