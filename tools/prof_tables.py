#!/usr/bin/env python3
"""cProfile of predict_tables on the bench_levels table (400 k genes / 600 k rows)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gecco_amd import predict, tables  # noqa: E402
from gecco_amd.crf import ClusterCRF  # noqa: E402

crf = ClusterCRF.trained(os.path.join(ROOT, "tests", "golden"))
attrs = crf.model.attributes_
rng = np.random.default_rng(0)
nc, per = 2000, 200
ng = nc * per
k = rng.integers(0, 4, size=ng)
owner = np.repeat(np.arange(ng), k)
nf = len(owner)
g_sid = np.array([f"contig_{c:05d}" for c in range(nc)], dtype=object)[np.arange(ng) // per]
g_pid = np.array([f"g{i:07d}" for i in range(ng)], dtype=object)
g_start = (np.arange(ng) % per) * 1000
genes_t = tables.GeneTable({"sequence_id": g_sid, "protein_id": g_pid, "start": g_start, "end": g_start + 900,
                            "strand": np.full(ng, "+", dtype=object)})
doms = np.array(attrs, dtype=object)[rng.integers(0, len(attrs), size=nf)]
feats_t = tables.FeatureTable({
    "sequence_id": g_sid[owner], "protein_id": g_pid[owner], "start": g_start[owner], "end": g_start[owner] + 900,
    "strand": np.full(nf, "+", dtype=object), "domain": doms, "hmm": np.full(nf, "Pfam", dtype=object),
    "i_evalue": np.full(nf, 1e-10), "pvalue": np.full(nf, 1e-12), "domain_start": rng.integers(1, 300, size=nf),
    "domain_end": np.full(nf, 300)})
predict.predict_tables(genes_t, feats_t, crf)
for _ in range(3):
    t0 = time.perf_counter()
    predict.predict_tables(genes_t, feats_t, crf)
    print("predict_tables ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile()
pr.enable()
predict.predict_tables(genes_t, feats_t, crf)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
