"""The reporting levels SURVEY.md §8d asks for, measured on one device: what a caller gets at every layer above the
resident-input kernel step -- host buffers through the C ABI (H2D + kernel + D2H), cluster calls only, table columns
through the columnar packer, `Gene` objects through the drop-in class.  Used by `bench.py` (a few iterations each,
after its timed region) and by `tools/bench_levels.py` (the longer sweep)."""
import time
import warnings

import numpy as np

from gecco_amd import _native as nat


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
    out.update(cluster_call_levels(ses, model, wl, cp, gp, at, at16, deg, outp, reps, W))
    for v in out.values():
        v["genes"] = n
        v["devices"] = len(devices)
    return out


def cluster_call_levels(ses, model, wl, cp, gp, at, at16, deg, outp, reps=5, W=20):
    """Cluster calls (marginals + refiner on the device) on pinned buffers.  `cluster_calls_pinned` keeps the workload it has
    had since round 3 (rows + the probabilities of their genes, 32-bit indices, `annotated` uploaded); the compact wire
    formats and the rows-only calls have keys of their own."""
    out = {}
    n = int(wl["contig_ptr"][-1])
    ann = nat.pinned_copy((np.diff(wl["gene_ptr"]) > 0).astype(np.uint8))
    seg, _, seg_off, _ = ses.clusters(cp, gp, at, ann, W, want_p=False, want_seg_p=True, degree=deg)
    common = {"clusters": int(len(seg)), "genes_in_clusters": int(seg_off[-1])}
    dt = _timed(lambda: ses.clusters(cp, gp, at, ann, W, want_p=False), reps)
    out["cluster_calls_pinned"] = {"ms": dt * 1e3, "genes_per_s": n / dt, **common, "h2d_mb": ses.stats()["h2d_bytes"] / 1e6,
                                   "note": "marginals + refiner on the device, rows AND the probabilities of their genes come back "
                                           "(gecco_crf_session_clusters: row pointers, 32-bit attribute indices and `annotated` cross PCIe; "
                                           "a fresh output array per call) -- the workload this key has had since round 3"}
    # (`annotated` left to the degree bytes: here a gene is annotated iff it has a domain, which is what they say)
    dt = _timed(lambda: ses.clusters(cp, gp, at, None, W, want_p=False, want_seg_p=False, degree=deg), reps)
    out["cluster_rows_i32_pinned"] = {"ms": dt * 1e3, "genes_per_s": n / dt, **common, "h2d_mb": ses.stats()["h2d_bytes"] / 1e6,
                                      "note": "rows only; degree bytes + 32-bit attribute indices on the wire (gecco_crf_session_clusters_degrees)"}
    if at16 is not None:
        dt = _timed(lambda: ses.clusters(cp, gp, at16, None, W, want_p=False, want_seg_p=False, degree=deg), reps)
        out["cluster_rows_wire16_pinned"] = {
            "ms": dt * 1e3, "genes_per_s": n / dt, **common, "h2d_mb": ses.stats()["h2d_bytes"] / 1e6,
            "note": "rows only; a degree byte per gene and 16-bit attribute indices cross PCIe (gecco_crf_session_clusters_wire; a model "
                    "with at most 65536 attributes: 3.8 bytes per gene), row pointers and 32-bit indices are rebuilt on the device "
                    "(rounds 3-4 reported this under `cluster_calls_pinned`)"}
    dt = _timed(lambda: ses.clusters(cp, gp, at, ann, W, want_p=False, degree=deg, seg_p_out=outp), reps)
    out["cluster_calls_with_probabilities_pinned"] = {
        "ms": dt * 1e3, "genes_per_s": n / dt, **common,
        "note": "rows + the probabilities of the clusters' genes (what a cluster table needs of p), degree bytes on the wire: gathered "
                "on the device, downloaded into a pinned caller buffer once the host knows how many there are"}
    return out


def multi_entry_level(model, wl, devices, reps=5, W=20):
    """One-shot windowed marginals on pinned buffers through a session over `devices` (an entry per listed device, repeats
    allowed: every entry has its own streams, lanes and submitting thread), with the host's issue time per chunk."""
    n = int(wl["contig_ptr"][-1])
    ses = nat.Session(model, list(devices))
    cp, gp, at = nat.pinned_copy(wl["contig_ptr"]), nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
    outp = nat.pinned_empty(n, np.float64)
    dt = _timed(lambda: ses.windowed_marginals(cp, gp, at, W, out=outp), reps)
    st = ses.stats()
    return {"ms": dt * 1e3, "genes_per_s": n / dt, "genes": n, "entries": len(devices), "chunks": st["n_chunks"], "host_threads": st["host_threads"],
            "host_issue_us_per_chunk": st["host_issue_seconds"] * 1e6 / max(st["n_chunks"], 1),
            "host_plan_us_per_chunk": st["host_plan_seconds"] * 1e6 / max(st["n_chunks"], 1),
            "note": "gecco_crf_session_windowed, pinned buffers; host_issue = time the submitting threads spend in HIP API calls and "
                    "chunk layouts per chunk (a 2^19-gene chunk is ~90 us of PCIe on its device)"}


def cluster_levels_for(model, wl, devices=(0,), reps=5, W=20):
    """The cluster-call levels alone, for a second weight law (bench.py: SURVEY.md 8d's law puts nine genes in ten into a
    cluster; the 'genome' law -- few clusters -- is what a metagenome looks like)."""
    n = int(wl["contig_ptr"][-1])
    ses = nat.Session(model, list(devices))
    cp, gp, at = nat.pinned_copy(wl["contig_ptr"]), nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
    at16 = nat.pinned_copy(wl["attr_id"], np.uint16) if model.num_attrs <= 65536 else None
    deg = nat.pinned_copy(nat.degree_bytes(wl["gene_ptr"]))
    out = cluster_call_levels(ses, model, wl, cp, gp, at, at16, deg, nat.pinned_empty(n, np.float64), reps, W)
    for v in out.values():
        v["genes"] = n
        v["devices"] = len(devices)
    return out


def object_level(model_dir, n_contigs=250, per=200, seed=0):
    """`ClusterCRF.predict_probabilities` on `Gene` objects of the real model (sort + pack + score + new objects)."""
    from gecco_amd.crf import ClusterCRF
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand

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
        breakdown = object_breakdown(crf, genes)
    return {"genes": len(genes), "ms": dt * 1e3, "genes_per_s": len(genes) / dt, "breakdown_us_per_gene": breakdown,
            "note": "ClusterCRF.predict_probabilities: sort + pack Gene objects + one-shot ABI + new Gene/Domain objects; breakdown = "
                    "the same steps timed one by one (sort: the order check / sort by (source.id, start) + every gene's domain list -- one "
                    "native pass that also groups AND packs when the input is in order, as annotation pipelines emit it (group and "
                    "pack are then 0); group: itertools.groupby into contigs otherwise; pack: objects -> CSR (csrc/objpath.c); abi: the batch driver call, host buffers "
                    "in and out; clone: new Gene / Protein / Domain objects with probability and cluster weight)"}


def object_breakdown(crf, genes):
    """Where `ClusterCRF.predict_probabilities` spends its time on `genes`, step by step (microseconds per gene): the steps
    of gecco_amd/crf.py, which are the reference's (/root/reference/gecco/crf/__init__.py:199-273), timed separately."""
    import gc
    import itertools
    import operator

    from gecco_amd import packing

    from gecco_amd._objpath_loader import module as _objpath

    n = max(len(genes), 1)
    native = _objpath()
    t = [time.perf_counter()]
    got = native.sort_group(genes, operator.attrgetter("start"), crf.model._attr_index) if native is not None else None
    if got is not None:  # (input already in order: ONE native pass checks, sorts the domain lists that need it, groups and packs)
        gs, contigs, ip, ap, at = got
        batch = packing.PackedBatch(np.frombuffer(ip, dtype=np.int64), np.frombuffer(ap, dtype=np.int64), np.frombuffer(at, dtype=np.int32))
        t += [time.perf_counter()] * 3
    else:
        gs = sorted(genes, key=operator.attrgetter("source.id", "start"))
        for g in gs:
            g.protein.domains.sort(key=operator.attrgetter("start"))
        t.append(time.perf_counter())
        contigs = [list(g) for _, g in itertools.groupby(gs, key=operator.attrgetter("source.id"))]
        t.append(time.perf_counter())
        batch = packing.pack_contigs(contigs, crf.model._attr_index, crf.feature_type)
        t.append(time.perf_counter())
    label = crf.model.native.label_id("1")
    p = crf._score(batch, crf.window_size, crf.window_step, label, True, lambda a, b: None, 0)
    t.append(time.perf_counter())
    out = []
    was = gc.isenabled()
    gc.disable()
    try:
        crf._annotate_contigs(contigs, np.ones(len(contigs), dtype=bool), batch, p, crf.model.state_features_, crf.model.cluster_weights_, out,
                              gs if isinstance(gs, list) else None)
    finally:
        if was:
            gc.enable()
    t.append(time.perf_counter())
    names = ("sort", "group", "pack", "abi", "clone")  # (one native pass: all of sort + group + pack is under "sort")
    return {k: (b - a) * 1e6 / n for k, a, b in zip(names, t[:-1], t[1:])}


def tables_level(model_dir, nc=1000, per=200, seed=0, reps=5):
    """`predict.predict_tables`: feature / gene table columns -> CSR -> device -> output columns + cluster rows."""
    from gecco_amd import predict, tables
    from gecco_amd.crf import ClusterCRF

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


# columns of clusters.tsv the CRF / refiner decide (`type` and the `*_probability` columns come from the type classifier's
# random forest: out of scope)
_CLUSTER_COLUMNS = ("sequence_id", "cluster_id", "start", "end", "average_p", "max_p", "proteins", "domains")
_FLOAT_COLUMNS = {"average_p", "max_p", "cluster_probability"}


def golden_table_identity(golden_dir, out_dir=None, reference_bits=False):
    """The reference's own acceptance test (/root/reference/galaxy/gecco.xml:83-111 asserts whole-file equality of
    genes.tsv / clusters.tsv): write the three tables of the BGC0001866 fixture through `python -m gecco_amd.predict` and
    count the cells that are not STRING-identical to the fixture's -- probabilities are printed with repr()'s 16-17 digits,
    so a cell differs as soon as a value is one ulp off.  Returns counts per table for text / integer cells and float cells,
    and the largest ulp distance among the float cells."""
    import csv
    import os
    import tempfile

    from gecco_amd import predict

    def rows(path):
        with open(path) as fh:
            return list(csv.DictReader(fh, delimiter="\t"))

    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        tmp = out_dir or tmp
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            predict.main(["--genes", os.path.join(golden_dir, "BGC0001866.genes.tsv"), "--features",
                          os.path.join(golden_dir, "BGC0001866.features.tsv"), "--model", golden_dir, "-o", tmp]
                         + (["--reference-bits"] if reference_bits else ["--fast-kernels"]))
        for table in ("genes", "features", "clusters"):
            got, ref = rows(os.path.join(tmp, f"BGC0001866.{table}.tsv")), rows(os.path.join(golden_dir, f"BGC0001866.{table}.tsv"))
            cols = _CLUSTER_COLUMNS if table == "clusters" else tuple(ref[0].keys())
            t = {"rows": len(got), "rows_expected": len(ref), "exact_cells": 0, "exact_cells_differing": 0, "float_cells": 0,
                 "float_cells_differing": 0, "max_ulps": 0}
            for a, b in zip(got, ref):
                for c in cols:
                    if table == "clusters" and c in ("proteins", "domains"):
                        # the fixture file predates the reference's current code for these two cells (it lists the proteins in
                        # gene order and every domain once; gecco/model.py:750-757 now sorts the ids as strings and lists every
                        # domain hit): compared with the CURRENT formula, evaluated on the fixture's own tables
                        t["formula_cells"] = t.get("formula_cells", 0) + 1
                        members = [r for r in rows(os.path.join(golden_dir, "BGC0001866.genes.tsv"))
                                   if r["sequence_id"] == b["sequence_id"] and int(b["start"]) <= int(r["start"]) and int(r["end"]) <= int(b["end"])]
                        ids = {r["protein_id"] for r in members}
                        if c == "proteins":
                            want = ";".join(sorted(ids))
                        else:
                            want = ";".join(sorted(r["domain"] for r in rows(os.path.join(golden_dir, "BGC0001866.features.tsv")) if r["protein_id"] in ids))
                        t["formula_cells_differing"] = t.get("formula_cells_differing", 0) + int(a[c] != want)
                        t["stale_fixture_cells"] = t.get("stale_fixture_cells", 0) + int(want != b[c])
                        continue
                    if c in _FLOAT_COLUMNS:
                        t["float_cells"] += 1
                        if a[c] != b[c]:
                            t["float_cells_differing"] += 1
                            x, y = np.float64(a[c]), np.float64(b[c])
                            t["max_ulps"] = max(t["max_ulps"], int(abs(int(x.view(np.int64)) - int(y.view(np.int64)))))
                    else:
                        t["exact_cells"] += 1
                        t["exact_cells_differing"] += int(a[c] != b[c])
            res[table] = t
    res["mode"] = "reference bits (CRFsuite's operation order, correctly rounded exp)" if reference_bits else "fast kernels"
    res["note"] = ("cells of the tables written by `python -m gecco_amd.predict` on the BGC0001866 fixture that are not string-identical "
                   "to the reference's own output files (tests/golden = /root/reference/tests/test_cli/data): text and integer "
                   "columns must be identical; float columns differ where the probability is an ulp or two away from CRFsuite's "
                   "(north star: 1e-6)")
    return res
