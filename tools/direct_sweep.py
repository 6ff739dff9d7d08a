#!/usr/bin/env python3
"""Where does the batch driver's direct path stop paying?  One-shot calls (pinned buffers) at a range of sizes with the path
forced on and forced off: windowed marginals, decode (compact wire format), cluster rows.  usage: tools/direct_sweep.py [sizes...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gecco_amd import _native as nat  # noqa: E402
from benchkit import latency  # noqa: E402


def med(fn, reps=200):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[reps // 2] * 1e6


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [2000, 5000, 10000, 20000, 40000, 65000, 100000, 200000]
    model = nat.Model.from_lcrf(latency.real_blob())
    out = {}
    for n in sizes:
        cptr, gptr, attr = latency.c1_batch(n, model.num_attrs)
        b = latency._Pinned(cptr, gptr, attr)
        row = {}
        for mode, genes in (("direct", 1 << 22), ("chunked", 0)):
            ses = nat.Session(model, [0])
            ses.set_chunk_genes(1 << 22)
            ses.set_direct_genes(genes)
            row[mode] = {
                "windowed_us": med(lambda: ses.windowed_marginals(b.cp, b.gp, b.at, 20, out=b.p)),
                "decode_wire_us": med(lambda: ses.decode(b.cp, b.gp, b.at16, 20, out_p=b.p, out_y=b.y, degree=b.deg)),
                "clusters_wire_us": med(lambda: ses.clusters(b.cp, b.gp, b.at16, None, 20, want_p=False, want_seg_p=False, degree=b.deg)),
            }
            assert ses.stats()["direct"] == (1 if genes else 0)
        out[str(n)] = row
        print(n, json.dumps(row), file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
