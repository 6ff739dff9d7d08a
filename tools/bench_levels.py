#!/usr/bin/env python3
"""Throughput of the windowed-marginals path at three levels:
kernel only (resident inputs) / one-shot C ABI with host buffers (H2D + kernel + D2H) /
Python object API (ClusterCRF.predict_probabilities on Gene objects, incl. host packing) /
columnar tables API (gecco_amd.predict: TSV columns -> CSR -> device -> TSV columns).
Run on the GPU box; prints one JSON object."""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from gecco_amd import _native as nat, synth  # noqa: E402
from gecco_amd.crf import ClusterCRF  # noqa: E402
from gecco_amd.model import Domain, Gene, Protein, Source, Strand  # noqa: E402


def main():
    out = {}
    wl = synth.workload("C3")
    n = int(wl["contig_ptr"][-1])
    model = nat.Model.from_tables(wl["w"], wl["trans"])
    dev = torch.device("cuda", 0)
    plan = nat.Plan(model, wl["contig_ptr"], 20, 1, True, device=0)
    gp = torch.from_numpy(wl["gene_ptr"]).to(dev)
    at = torch.from_numpy(wl["attr_id"]).to(dev)
    p = torch.zeros(n, dtype=torch.float64, device=dev)
    ms = plan.time_windowed(gp.data_ptr(), at.data_ptr(), p.data_ptr(), 1, 0, warmup=3, iters=20)
    out["kernel_only"] = {"genes": n, "ms": ms, "genes_per_s": n / ms * 1e3}
    # one-shot ABI: chunk layouts + H2D + kernel + D2H per call, host numpy (pageable) buffers, on the model's own session
    def timed(fn, reps=10):
        fn()
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps

    dt = timed(lambda: model.windowed_marginals(wl["contig_ptr"], wl["gene_ptr"], wl["attr_id"], 20))
    out["one_shot_host_buffers"] = {"genes": n, "ms": dt * 1e3, "genes_per_s": n / dt,
                                    "note": "pageable numpy buffers: chunk layouts + H2D + kernel + D2H per call"}
    # the same through an explicit session with pinned buffers (gecco_crf_host_alloc): every copy asynchronous,
    # chunks pipelined; and the cluster-call variant, where the probabilities never leave the device
    ses = nat.Session(model, [0])
    cp, gpp, atp = nat.pinned_copy(wl["contig_ptr"]), nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
    outp = nat.pinned_empty(n, np.float64)
    ann = nat.pinned_copy((np.diff(wl["gene_ptr"]) > 0).astype(np.uint8))
    for chunk in (1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 22):
        ses.set_chunk_genes(chunk)
        dt = timed(lambda: ses.windowed_marginals(cp, gpp, atp, 20, out=outp))
        st = ses.stats()
        out[f"one_shot_pinned_chunk_{chunk}"] = {"genes": n, "ms": dt * 1e3, "genes_per_s": n / dt, "chunks": st["n_chunks"],
                                                 "host_plan_ms": st["host_plan_seconds"] * 1e3,
                                                 "h2d_mb": st["h2d_bytes"] / 1e6, "d2h_mb": st["d2h_bytes"] / 1e6}
    best = min((k for k in out if k.startswith("one_shot_pinned_chunk_")), key=lambda k: out[k]["ms"])
    out["one_shot_pinned"] = dict(out[best], note=f"best chunk size: {best.rsplit('_', 1)[1]} genes")
    ses.set_chunk_genes(int(best.rsplit("_", 1)[1]))
    dt = timed(lambda: ses.clusters(cp, gpp, atp, ann, 20, want_p=False, want_seg_p=True))
    seg = ses.clusters(cp, gpp, atp, ann, 20, want_p=False, want_seg_p=True)[0]
    out["one_shot_pinned_cluster_calls"] = {"genes": n, "ms": dt * 1e3, "genes_per_s": n / dt, "clusters": len(seg),
                                            "note": "marginals + refiner on the device, rows and their probabilities come back "
                                                    "into fresh pageable arrays (32-bit wire format)"}
    # the levels of the bench line (pinned buffers in and out, degree bytes / 16-bit indices on the wire, decode, cluster calls)
    from benchkit import levels  # noqa: E402
    ses.set_chunk_genes(1 << 19)
    out["batch_driver_levels"] = levels.host_buffer_levels(model, wl, reps=10)
    # object API on the real model: 500 contigs x 200 genes of Gene objects
    golden = os.path.join(ROOT, "tests", "golden")
    crf = ClusterCRF.trained(golden)
    attrs = crf.model.attributes_
    rng = np.random.default_rng(0)
    genes = []
    for c in range(500):
        src = Source(f"contig_{c:04d}")
        for i in range(200):
            k = int(rng.integers(0, 4))
            doms = [Domain(attrs[a], 10 * j + 1, 10 * j + 9, "Pfam", 1e-10, 1e-12) for j, a in enumerate(rng.integers(0, len(attrs), size=k))]
            genes.append(Gene(src, 1000 * i, 1000 * i + 900, Strand.Coding, Protein(f"c{c:04d}_{i}", None, doms)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        crf.predict_probabilities(genes[:2000])  # warm
        t0 = time.perf_counter()
        crf.predict_probabilities(genes)
        dt = time.perf_counter() - t0
    out["python_object_api"] = {"genes": len(genes), "ms": dt * 1e3, "genes_per_s": len(genes) / dt,
                                "note": "sort + pack Gene objects + one-shot ABI + new Gene/Domain objects"}
    # columnar path: the same kind of data as feature / gene tables (2000 contigs x 200 genes)
    import io
    import tempfile

    from gecco_amd import predict, tables

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
    predict.predict_tables(genes_t, feats_t, crf)  # warm: text columns go to Arrow layout once, buffers get sized
    predict.predict_tables(genes_t, feats_t, crf)
    # (the object level above leaves half a million Python objects behind: a full pass of the cyclic collector inside a timed call
    # costs ~9 ms.  The table path itself creates a few hundred objects: collected before, suspended during.  Sporadic calls of
    # 20-30 ms remain on some boxes -- the median of 9 ignores them)
    import gc

    gc.collect()
    gc.disable()
    ts = []
    try:
        for _ in range(9):
            t0 = time.perf_counter()
            g_out, f_out, c_out = predict.predict_tables(genes_t, feats_t, crf)
            ts.append(time.perf_counter() - t0)
    finally:
        gc.enable()
    dt = sorted(ts)[len(ts) // 2]
    out["columnar_tables_api"] = {"genes": ng, "domain_rows": nf, "clusters": len(c_out), "ms": dt * 1e3, "genes_per_s": ng / dt,
                                  "ms_all": [round(t * 1e3, 2) for t in ts],
                                  "note": "median of 9: native packer (table columns -> CSR in pinned memory) + batch driver "
                                          "(marginals + refiner on the device) + native cluster rows + output columns"}
    with tempfile.TemporaryDirectory() as tmp:
        gp_, fp_ = os.path.join(tmp, "x.genes.tsv"), os.path.join(tmp, "x.features.tsv")
        genes_t.dump(gp_)
        feats_t.dump(fp_)
        t0 = time.perf_counter()
        rc = predict.main(["--genes", gp_, "--features", fp_, "--model", golden, "-o", os.path.join(tmp, "out")])
        dt = time.perf_counter() - t0
    out["columnar_cli_tsv_to_tsv"] = {"genes": ng, "ms": dt * 1e3, "genes_per_s": ng / dt,
                                      "note": "python -m gecco_amd.predict: read 2 TSVs, predict, write 3 TSVs (includes model load)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
