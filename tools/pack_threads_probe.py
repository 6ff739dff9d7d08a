import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["GECCO_AMD_HIP_RUNTIME"] = "system"
import numpy as np
from gecco_amd import packing, tables, pickle_model, _native as nat
from benchkit.latency import real_blob
model = nat.Model.from_lcrf(real_blob())
attrs = model.attrs()
rng = np.random.default_rng(0)
nc, per = 2000, 200
ng = nc * per
k = rng.integers(0, 4, size=ng)
owner = np.repeat(np.arange(ng), k)
nf = len(owner)
g_sid = np.array([f"contig_{c:05d}" for c in range(nc)], dtype=object)[np.arange(ng) // per]
g_pid = np.array([f"g{i:07d}" for i in range(ng)], dtype=object)
g_start = (np.arange(ng) % per) * 1000
genes_t = tables.GeneTable({"sequence_id": g_sid, "protein_id": g_pid, "start": g_start, "end": g_start + 900, "strand": np.full(ng, "+", dtype=object)})
doms = np.array(attrs, dtype=object)[rng.integers(0, len(attrs), size=nf)]
feats_t = tables.FeatureTable({"sequence_id": g_sid[owner], "protein_id": g_pid[owner], "start": g_start[owner], "end": g_start[owner] + 900,
    "strand": np.full(nf, "+", dtype=object), "domain": doms, "hmm": np.full(nf, "Pfam", dtype=object), "i_evalue": np.full(nf, 1e-10),
    "pvalue": np.full(nf, 1e-12), "domain_start": rng.integers(1, 300, size=nf), "domain_end": np.full(nf, 300)})
packing.pack_tables(model, feats_t, genes_t)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); packing.pack_tables(model, feats_t, genes_t); ts.append(time.perf_counter() - t0)
print(os.environ.get("GECCO_CRF_HOST_THREADS"), "pack ms", round(sorted(ts)[3] * 1e3, 2), "genes", ng, "rows", nf)
