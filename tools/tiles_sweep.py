#!/usr/bin/env python3
"""One window tile per workgroup against two, over batch sizes (contigs of 200 genes, C2's model law): window kernel alone and
the pipelined decode launch on one stream, us.  Each setting runs in its own process (the limit is read once).
    python tools/tiles_sweep.py            (on the GPU box)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIZES = (20_000, 50_000, 100_000, 200_000, 300_000, 400_000, 600_000, 800_000, 1_200_000, 2_000_000)


def child():
    import numpy as np
    import torch

    from gecco_amd import _native as nat, synth

    wl = synth.workload("C2")
    model = nat.Model.from_tables(wl["w"], wl["trans"])
    dev = torch.device("cuda", 0)
    res = {}
    for n in SIZES:
        rng = np.random.default_rng(n)
        cptr, gptr, attr = synth.synth_contigs(rng, [200] * (n // 200), model.num_attrs)
        plan = nat.Plan(model, cptr, 20, 1, True, device=0)
        gp, at = torch.from_numpy(gptr).to(dev), torch.from_numpy(attr).to(dev)
        p = torch.zeros(n, dtype=torch.float64, device=dev)
        y = torch.zeros(n + 64, dtype=torch.int8, device=dev)
        w = plan.time_windowed(gp.data_ptr(), at.data_ptr(), p.data_ptr(), 1, 0, warmup=5, iters=100)
        d = plan.time_decode_pipelined(gp.data_ptr(), at.data_ptr(), p.data_ptr(), y.data_ptr(), 1, 0, warmup=5, iters=100)
        res[n] = (round(w * 1e3, 2), round(d * 1e3, 2))
    print("RES", json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    table = {}
    for rep in range(2):
        for name, limit in (("two", "0"), ("one", "2000000000")):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True,
                                 env=dict(os.environ, GECCO_CRF_TILES1_MAX_SLOTS=limit))
            line = [l for l in out.stdout.splitlines() if l.startswith("RES ")]
            if not line:
                print(out.stderr[-600:])
                continue
            table[(name, rep)] = json.loads(line[-1][4:])
    print("genes      window kernel two / one (us)      pipelined launch two / one (us)")
    for n in SIZES:
        row = [table.get((k, r), {}).get(str(n), (None, None)) for k in ("two", "one") for r in range(2)]
        print(f"{n:9d}  {row[0][0]} {row[1][0]} / {row[2][0]} {row[3][0]}      {row[0][1]} {row[1][1]} / {row[2][1]} {row[3][1]}")
