#!/usr/bin/env python3
"""Timeline of one fused decode launch (crf_decode_fused) from the per-block wall-clock stamps a -DGECCO_FUSED_TRACE
variant of the library writes (GECCO_CRF_FUSED_TRACE=<file>): when window tiles and Viterbi workgroups start, when the
Viterbi workgroups see their flags, when they end.  usage: fused_trace.py <file>"""
import sys

import numpy as np


def main():
    raw = open(sys.argv[1], "rb").read()
    nb, nt = np.frombuffer(raw[:16], dtype=np.int64)
    role = np.frombuffer(raw[16:16 + 4 * nb], dtype=np.int32)
    tr = np.frombuffer(raw[16 + 4 * nb:16 + 4 * nb + 32 * nb], dtype=np.uint64).reshape(nb, 4).astype(np.float64)
    t0 = tr[tr[:, 0] > 0, 0].min()
    us = lambda x: (x - t0) / 100.0  # 100 MHz wall clock
    tiles = role >= 0
    vd = (role < 0) & (role != np.iinfo(np.int32).min)
    print(f"blocks {nb}, tiles {tiles.sum()}, viterbi workgroups {vd.sum()}")
    for name, m in (("tiles", tiles), ("viterbi", vd)):
        s, e = us(tr[m, 0]), us(tr[m, 2])
        q = [0, 10, 50, 90, 100]
        print(f"{name:8s} start pct{q}: {np.percentile(s, q).round(1)}  end: {np.percentile(e, q).round(1)}  life: {np.percentile(e - s, q).round(1)}")
    s, f, e = us(tr[vd, 0]), us(tr[vd, 1]), us(tr[vd, 2])
    print("viterbi wait for flags:", np.percentile(f - s, [0, 10, 50, 90, 100]).round(1), " body:", np.percentile(e - f, [0, 10, 50, 90, 100]).round(1))
    # occupancy over time: blocks resident per 1-us bin
    end = us(tr[tiles | vd, 2]).max()
    bins = np.arange(0, end + 1.0, 2.0)
    for name, m in (("tiles", tiles), ("viterbi", vd)):
        s_, e_ = us(tr[m, 0]), us(tr[m, 2])
        occ = [int(((s_ <= b) & (e_ > b)).sum()) for b in bins]
        print(f"resident {name:8s} every 2 us:", occ)
    print("launch span:", us(tr[tiles | vd, 2]).max().round(1), "us; last tile end", us(tr[tiles, 2]).max().round(1))


if __name__ == "__main__":
    main()
