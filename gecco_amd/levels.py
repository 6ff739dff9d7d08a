"""The reporting levels SURVEY.md §8d asks for, measured on one device: what a caller gets at every layer above the
resident-input kernel step -- host buffers through the C ABI (H2D + kernel + D2H), cluster calls only, table columns
through the columnar packer, `Gene` objects through the drop-in class.  Used by `bench.py` (a few iterations each,
after its timed region) and by `tools/bench_levels.py` (the longer sweep)."""
import time
import warnings

import numpy as np

from . import _native as nat


def _timed(fn, reps):
    """median seconds per call (a single hiccup of the box -- a few ms -- would otherwise be a fifth of a five-call mean)"""
    fn()
    fn()
    ts = []
    for _ in range(max(reps, 3)):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def host_buffer_levels(model, wl, devices=(0,), reps=5, W=20):
    """One-shot calls on host buffers: pageable numpy arrays, pinned buffers, cluster calls only (pinned)."""
    out = {}
    n = int(wl["contig_ptr"][-1])
    ses = nat.Session(model, list(devices))
    dt = _timed(lambda: ses.windowed_marginals(wl["contig_ptr"], wl["gene_ptr"], wl["attr_id"], W), reps)
    out["one_shot_pageable"] = {"ms": dt * 1e3, "genes_per_s": n / dt,
                                "note": "pageable numpy buffers: chunk layouts + H2D + kernel + D2H per call"}
    cp, gp, at = nat.pinned_copy(wl["contig_ptr"]), nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
    outp = nat.pinned_empty(n, np.float64)
    dt = _timed(lambda: ses.windowed_marginals(cp, gp, at, W, out=outp), reps)
    st = ses.stats()
    out["one_shot_pinned"] = {"ms": dt * 1e3, "genes_per_s": n / dt, "chunks": st["n_chunks"], "h2d_mb": st["h2d_bytes"] / 1e6,
                              "d2h_mb": st["d2h_bytes"] / 1e6,
                              "note": "pinned buffers (gecco_crf_host_alloc): every copy asynchronous, chunks pipelined"}
    deg = nat.pinned_copy(nat.degree_bytes(wl["gene_ptr"]))
    dt = _timed(lambda: ses.windowed_marginals(cp, gp, at, W, out=outp, degree=deg), reps)
    st = ses.stats()
    out["one_shot_pinned_degree_bytes"] = {"ms": dt * 1e3, "genes_per_s": n / dt, "h2d_mb": st["h2d_bytes"] / 1e6,
                                           "note": "the same with one degree byte per gene on the wire instead of a 4-byte row pointer "
                                                   "(gecco_crf_session_windowed_degrees); row pointers rebuilt on the device"}
    outy = nat.pinned_empty(n, np.int8)
    dt = _timed(lambda: ses.decode(cp, gp, at, W, out_p=outp, out_y=outy), reps)
    st = ses.stats()
    out["decode_pinned"] = {"ms": dt * 1e3, "genes_per_s": n / dt, "chunks": st["n_chunks"], "h2d_mb": st["h2d_bytes"] / 1e6,
                            "d2h_mb": st["d2h_bytes"] / 1e6,
                            "note": "the metric's own step at this level (gecco_crf_session_decode): windowed marginals + Viterbi labels, "
                                    "pinned buffers in and out; one pipelined launch per chunk (window tiles + the Viterbi workgroups "
                                    "of the chunk before), a flush at the end"}
    at16 = nat.pinned_copy(wl["attr_id"], np.uint16) if model.num_attrs <= 65536 else None
    if at16 is not None:
        dt = _timed(lambda: ses.decode(cp, gp, at16, W, out_p=outp, out_y=outy, degree=deg), reps)
        st = ses.stats()
        out["decode_pinned_wire16"] = {
            "ms": dt * 1e3, "genes_per_s": n / dt, "chunks": st["n_chunks"], "h2d_mb": st["h2d_bytes"] / 1e6, "d2h_mb": st["d2h_bytes"] / 1e6,
            "note": "the same with the compact wire format (gecco_crf_session_decode_wire): a degree byte per gene and 16-bit "
                    "attribute indices cross PCIe, row pointers and 32-bit indices are rebuilt on the device"}
    ann = nat.pinned_copy((np.diff(wl["gene_ptr"]) > 0).astype(np.uint8))
    # (`annotated` left to the degree bytes: here a gene is annotated iff it has a domain, which is what they say)
    dt = _timed(lambda: ses.clusters(cp, gp, at, None, W, want_p=False, want_seg_p=False, degree=deg), reps)
    h2d = ses.stats()["h2d_bytes"]
    seg, _, seg_off, _ = ses.clusters(cp, gp, at, ann, W, want_p=False, want_seg_p=True, degree=deg)
    rows32 = {"ms": dt * 1e3, "genes_per_s": n / dt, "clusters": int(len(seg)), "genes_in_clusters": int(seg_off[-1]), "h2d_mb": h2d / 1e6,
              "note": "marginals + refiner on the device, degree bytes + 32-bit attribute indices on the wire "
                      "(gecco_crf_session_clusters_degrees); only the cluster rows come back"}
    if at16 is not None:
        # the level's own entry is the compact wire format (what a caller in a hurry sends); the 32-bit one sits beside it
        dt = _timed(lambda: ses.clusters(cp, gp, at16, None, W, want_p=False, want_seg_p=False, degree=deg), reps)
        out["cluster_calls_pinned"] = {
            "ms": dt * 1e3, "genes_per_s": n / dt, "clusters": int(len(seg)), "genes_in_clusters": int(seg_off[-1]),
            "h2d_mb": ses.stats()["h2d_bytes"] / 1e6,
            "note": "marginals + refiner on the device; a degree byte per gene and 16-bit attribute indices cross PCIe "
                    "(gecco_crf_session_clusters_wire; a model with at most 65536 attributes: 3.8 bytes per gene), row pointers "
                    "and 32-bit indices are rebuilt on the device; only the cluster rows come back"}
        out["cluster_calls_pinned_i32"] = rows32
    else:
        out["cluster_calls_pinned"] = rows32
    dt = _timed(lambda: ses.clusters(cp, gp, at, ann, W, want_p=False, degree=deg, seg_p_out=outp), reps)
    out["cluster_calls_with_probabilities_pinned"] = {
        "ms": dt * 1e3, "genes_per_s": n / dt, "clusters": int(len(seg)), "genes_in_clusters": int(seg_off[-1]),
        "note": "the same + the probabilities of the clusters' genes (what a cluster table needs of p): gathered on the device, "
                "downloaded into a pinned caller buffer once the host knows how many there are"}
    for v in out.values():
        v["genes"] = n
        v["devices"] = len(devices)
    return out


def object_level(model_dir, n_contigs=250, per=200, seed=0):
    """`ClusterCRF.predict_probabilities` on `Gene` objects of the real model (sort + pack + score + new objects)."""
    from .crf import ClusterCRF
    from .model import Domain, Gene, Protein, Source, Strand

    crf = ClusterCRF.trained(model_dir)
    attrs = crf.model.attributes_
    rng = np.random.default_rng(seed)
    genes = []
    for c in range(n_contigs):
        src = Source(f"contig_{c:04d}")
        for i in range(per):
            k = int(rng.integers(0, 4))
            doms = [Domain(attrs[a], 10 * j + 1, 10 * j + 9, "Pfam", 1e-10, 1e-12)
                    for j, a in enumerate(rng.integers(0, len(attrs), size=k))]
            genes.append(Gene(src, 1000 * i, 1000 * i + 900, Strand.Coding, Protein(f"c{c:04d}_{i}", None, doms)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        crf.predict_probabilities(genes[:2000])  # warm
        t0 = time.perf_counter()
        crf.predict_probabilities(genes)
        dt = time.perf_counter() - t0
    return {"genes": len(genes), "ms": dt * 1e3, "genes_per_s": len(genes) / dt,
            "note": "ClusterCRF.predict_probabilities: sort + pack Gene objects + one-shot ABI + new Gene/Domain objects"}


def tables_level(model_dir, nc=1000, per=200, seed=0, reps=5):
    """`predict.predict_tables`: feature / gene table columns -> CSR -> device -> output columns + cluster rows."""
    from . import predict, tables
    from .crf import ClusterCRF

    crf = ClusterCRF.trained(model_dir)
    attrs = crf.model.attributes_
    rng = np.random.default_rng(seed)
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
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _, _, c_out = predict.predict_tables(genes_t, feats_t, crf)
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    return {"genes": ng, "domain_rows": int(nf), "clusters": int(len(c_out)), "ms": dt * 1e3, "genes_per_s": ng / dt,
            "note": f"predict_tables, median of {reps}: native packer (table columns -> CSR in pinned memory) + batch driver "
                    "(marginals + refiner on the device) + native cluster rows + output columns"}
