#!/usr/bin/env python3
"""Session over k entries of device 0 on C3: wall, issue time per chunk; GECCO_CRF_TRACE=1 shows the host time of every step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from gecco_amd import _native as nat, synth

wl = synth.workload("C3")
model = nat.Model.from_tables(wl["w"], wl["trans"])
n = int(wl["contig_ptr"][-1])
cp, gp, at = nat.pinned_copy(wl["contig_ptr"]), nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
p = nat.pinned_empty(n, np.float64)
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    ses = nat.Session(model, [0] * k)
    for _ in range(3):
        ses.windowed_marginals(cp, gp, at, 20, out=p)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); ses.windowed_marginals(cp, gp, at, 20, out=p); ts.append(time.perf_counter() - t0)
    st = ses.stats()
    print(k, "entries: wall %.3f ms" % (sorted(ts)[4] * 1e3), "chunks", st["n_chunks"], "issue/chunk %.1f us" % (st["host_issue_seconds"] * 1e6 / st["n_chunks"]),
          "plan/chunk %.1f us" % (st["host_plan_seconds"] * 1e6 / st["n_chunks"]), flush=True)
    if os.environ.get("PROBE_TRACE") == str(k):
        os.environ["GECCO_CRF_TRACE"] = "1"
