"""GPU end-to-end: the drop-in ``ClusterCRF`` (real HIP engine, no stand-ins) reproduces the
reference's golden tables for ``gecco run`` on BGC0001866 (features -> genes -> clusters)."""
import os
import warnings

import numpy as np
import pytest

from tests.helpers import GOLDEN, read_tsv

pytestmark = pytest.mark.gpu


def test_golden_run_through_dropin_class():
    from gecco_amd import _native, refine
    from gecco_amd.crf import ClusterCRF
    from tests.test_host_logic import _golden_genes

    assert _native.device_count() >= 1
    crf = ClusterCRF.trained(GOLDEN)
    calls = []
    out = crf.predict_probabilities(_golden_genes(), progress=lambda i, t: calls.append((i, t)))
    rows = read_tsv(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    assert [g.protein.id for g in out] == [r["protein_id"] for r in rows]
    err = max(abs(g.average_probability - float(r["average_p"])) for g, r in zip(out, rows))
    assert err <= 1e-14, err
    assert calls[0] == (0, 4) and calls[-1] == (4, 4)
    clusters = list(refine.ClusterRefiner(threshold=0.8, n_cds=3).iter_clusters(out))
    row = read_tsv(os.path.join(GOLDEN, "BGC0001866.clusters.tsv"))[0]
    assert len(clusters) == 1
    c = clusters[0]
    assert (c.id, c.start, c.end) == (row["cluster_id"], int(row["start"]), int(row["end"]))
    assert abs(c.average_probability - float(row["average_p"])) <= 1e-14
    assert abs(c.maximum_probability - float(row["max_p"])) <= 1e-14


def test_many_contigs_objects_vs_oracle(oracle_model):
    """Object-level API on a multi-contig batch incl. short contigs, batched launches."""
    from gecco_amd.crf import ClusterCRF
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(3)
    attrs = oracle_model["attrs"]
    genes, cptr, gptr, attr = [], [0], [0], []
    for c in range(40):
        n = int(rng.integers(3, 120))
        for i in range(n):
            k = int(rng.integers(0, 4))
            ids = rng.integers(0, len(attrs), size=k)
            doms = [Domain(attrs[a], 10 * j + 1, 10 * j + 9, "Pfam", 1e-10, 1e-12) for j, a in enumerate(ids)]
            if rng.random() < 0.1:
                doms.append(Domain("PF99999", 500, 510, "Pfam", 1e-10, 1e-12))  # unknown to the model
            genes.append(Gene(Source(f"contig_{c:03d}"), 1000 * i, 1000 * i + 900, Strand.Coding, Protein(f"c{c:03d}_{i}", None, doms)))
            seen = []
            for a in ids.tolist():
                if a not in seen:
                    seen.append(a)
            attr += seen
            gptr.append(len(attr))
        cptr.append(cptr[-1] + n)
    crf = ClusterCRF.trained(GOLDEN)
    crf._BATCH_GENES = 500  # force several launches
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = crf.predict_probabilities(list(reversed(genes)), pad=True)
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, True)
    got = np.array([g.average_probability for g in out])
    assert np.abs(got - exp).max() <= 1e-12
    # columnar entry point gives the same numbers
    p = crf.predict_probabilities_csr(cptr, gptr, attr)
    assert np.abs(p - exp).max() <= 1e-12


def test_predict_clusters_convenience():
    from gecco_amd.crf import ClusterCRF
    from tests.test_host_logic import _golden_genes

    clusters = ClusterCRF.trained(GOLDEN).predict_clusters(_golden_genes())
    assert [c.id for c in clusters] == ["BGC0001866.1_cluster_1"] and len(clusters[0].genes) == 23


def test_columnar_predict_cli_reproduces_golden_tables(tmp_path):
    """python -m gecco_amd.predict on the reference's fixture inputs -> the reference's outputs."""
    from gecco_amd import predict

    rc = predict.main(["--genes", os.path.join(GOLDEN, "BGC0001866.genes.tsv"),
                       "--features", os.path.join(GOLDEN, "BGC0001866.features.tsv"),
                       "--model", GOLDEN, "-o", str(tmp_path)])
    assert rc == 0
    got = read_tsv(str(tmp_path / "BGC0001866.genes.tsv"))
    ref = read_tsv(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    assert [r["protein_id"] for r in got] == [r["protein_id"] for r in ref]
    for a, b in zip(got, ref):
        assert all(a[k] == b[k] for k in ("sequence_id", "start", "end", "strand"))
        assert abs(float(a["average_p"]) - float(b["average_p"])) <= 1e-14
        assert abs(float(a["max_p"]) - float(b["max_p"])) <= 1e-14
    got = read_tsv(str(tmp_path / "BGC0001866.features.tsv"))
    ref = read_tsv(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    assert len(got) == len(ref) == 37
    for a, b in zip(got, ref):
        assert all(a[k] == b[k] for k in ("protein_id", "domain", "hmm", "i_evalue", "pvalue", "domain_start", "domain_end"))
        assert abs(float(a["cluster_probability"]) - float(b["cluster_probability"])) <= 1e-14
    got = read_tsv(str(tmp_path / "BGC0001866.clusters.tsv"))
    ref = read_tsv(os.path.join(GOLDEN, "BGC0001866.clusters.tsv"))
    assert len(got) == len(ref) == 1
    a, b = got[0], ref[0]
    assert all(a[k] == b[k] for k in ("sequence_id", "cluster_id", "start", "end"))
    assert abs(float(a["average_p"]) - float(b["average_p"])) <= 1e-14 and abs(float(a["max_p"]) - float(b["max_p"])) <= 1e-14
    assert set(a["proteins"].split(";")) == set(b["proteins"].split(";"))
    assert set(a["domains"].split(";")) == set(b["domains"].split(";"))


def test_cli_tables_string_identity_with_the_reference_files(tmp_path, capsys):
    """The reference's own acceptance test (/root/reference/galaxy/gecco.xml:83-111: whole-file equality of genes.tsv and
    clusters.tsv): every text / integer cell written by `python -m gecco_amd.predict` is string-identical to the fixture's;
    float cells (probabilities printed with 16-17 digits) are counted and reported: a cell differs as soon as the value is one
    ulp away from CRFsuite's, and must never be more than 8 ulps away."""
    from benchkit import levels

    res = levels.golden_table_identity(GOLDEN, str(tmp_path))
    with capsys.disabled():
        print("\n[table identity]", {k: v for k, v in res.items() if k != "note"})
    for table, n_rows in (("genes", 23), ("features", 37), ("clusters", 1)):
        t = res[table]
        assert t["rows"] == t["rows_expected"] == n_rows
        assert t["exact_cells_differing"] == 0
        assert t["max_ulps"] <= 32  # (3.6e-15 at p ~ 1: the fast kernels' own summation order; north star 1e-6)
    assert res["genes"]["float_cells"] == 46 and res["clusters"]["float_cells"] == 2
    # `proteins` / `domains`: the reference's CURRENT formula (gecco/model.py:750-757) on the fixture's tables
    assert res["clusters"]["formula_cells"] == 2 and res["clusters"]["formula_cells_differing"] == 0


def test_columnar_predict_cli_postproc_antismash(tmp_path):
    """`--postproc antismash` on the fixture: the cluster is kept or dropped exactly as ClusterRefiner(criterion=
    "antismash", n_cds=3) decides on the fixture's objects (refine.py:157-163 with the CLI's defaults)."""
    from gecco_amd import predict, refine, tables

    rc = predict.main(["--genes", os.path.join(GOLDEN, "BGC0001866.genes.tsv"),
                       "--features", os.path.join(GOLDEN, "BGC0001866.features.tsv"),
                       "--model", GOLDEN, "--postproc", "antismash", "-o", str(tmp_path)])
    assert rc == 0
    got = read_tsv(str(tmp_path / "BGC0001866.clusters.tsv")) if os.path.exists(tmp_path / "BGC0001866.clusters.tsv") else []
    feats = tables.FeatureTable.load(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    genes_t = tables.GeneTable.load(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    by_pid = {g.protein.id: g for g in genes_t.to_genes()}
    for g in feats.to_genes():
        by_pid[g.protein.id].protein.domains.extend(g.protein.domains)
    exp = list(refine.ClusterRefiner(criterion="antismash", n_cds=3).iter_clusters(list(by_pid.values())))
    n_markers = len({d.name for g in by_pid.values() for d in g.protein.domains} & refine.BIO_PFAMS)
    assert len(got) == len(exp) == (1 if n_markers >= 5 else 0)
    if exp:
        assert got[0]["cluster_id"] == exp[0].id and len(got[0]["proteins"].split(";")) == len(exp[0].genes)


def test_multi_device_sharding_code_path(oracle_model):
    """ClusterCRF.devices with more than one entry shards launches over a thread pool; exercised
    here with the same physical device twice."""
    from gecco_amd.crf import ClusterCRF
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand

    rng = np.random.default_rng(8)
    attrs = oracle_model["attrs"]
    genes = []
    for c in range(12):
        for i in range(int(rng.integers(25, 90))):
            doms = [Domain(attrs[a], 1, 9, "Pfam", 1e-10, 1e-12) for a in rng.integers(0, len(attrs), size=int(rng.integers(0, 3)))]
            genes.append(Gene(Source(f"k{c:02d}"), 100 * i, 100 * i + 90, Strand.Coding, Protein(f"k{c:02d}_{i}", None, doms)))
    one = ClusterCRF.trained(GOLDEN)
    two = ClusterCRF.trained(GOLDEN)
    two.devices = [0, 0]
    two._BATCH_GENES = 150
    a = [g.average_probability for g in one.predict_probabilities(list(genes))]
    b = [g.average_probability for g in two.predict_probabilities(list(genes))]
    assert a == b


def _random_tables(rng, n_contigs, attrs, shuffle):
    """Gene + feature tables of `n_contigs` contigs (lengths 3..120: some shorter than the window)."""
    from gecco_amd import tables

    g_sid, g_pid, g_start, g_end = [], [], [], []
    f_rows = []
    for c in range(n_contigs):
        n = int(rng.integers(3, 120))
        pos = 0
        for i in range(n):
            pos += int(rng.integers(50, 900))
            g_sid.append(f"ctg{c:03d}")
            g_pid.append(f"ctg{c:03d}_{i}")
            g_start.append(pos)
            g_end.append(pos + int(rng.integers(60, 800)))
            for _ in range(int(rng.integers(0, 4))):
                f_rows.append((g_sid[-1], g_pid[-1], g_start[-1], g_end[-1], attrs[int(rng.integers(0, len(attrs)))],
                               int(rng.integers(1, 200))))
    gi = np.arange(len(g_pid))
    fi = np.arange(len(f_rows))
    if shuffle:
        rng.shuffle(gi)
        rng.shuffle(fi)
    genes_t = tables.GeneTable({"sequence_id": np.array(g_sid, dtype=object)[gi], "protein_id": np.array(g_pid, dtype=object)[gi],
                                "start": np.array(g_start)[gi], "end": np.array(g_end)[gi],
                                "strand": np.full(len(gi), "+", dtype=object)})
    fr = [f_rows[i] for i in fi]
    feats_t = tables.FeatureTable({
        "sequence_id": np.array([r[0] for r in fr], dtype=object), "protein_id": np.array([r[1] for r in fr], dtype=object),
        "start": np.array([r[2] for r in fr]), "end": np.array([r[3] for r in fr]), "strand": np.full(len(fr), "+", dtype=object),
        "domain": np.array([r[4] for r in fr], dtype=object), "hmm": np.full(len(fr), "Pfam", dtype=object),
        "i_evalue": np.full(len(fr), 1e-10), "pvalue": np.full(len(fr), 1e-12),
        "domain_start": np.array([r[5] for r in fr]), "domain_end": np.array([r[5] + 10 for r in fr])})
    return genes_t, feats_t


@pytest.mark.parametrize("shuffle,pad", [(False, True), (True, True), (True, False)])
def test_columnar_predict_equals_object_path(oracle_model, shuffle, pad):
    """predict_tables (native packer -> batch driver -> resident refiner -> native cluster rows) against the
    object path (predict_probabilities + ClusterRefiner per contig, i.e. what `gecco run` does with Gene
    objects) on random tables, in order and shuffled, with and without padding."""
    _columnar_against_objects(oracle_model, 21, 40, shuffle, pad, 0.3, 2, min_clusters=6)


@pytest.mark.parametrize("seed", range(12))
def test_columnar_predict_equals_object_path_random(oracle_model, seed):
    """The same equivalence over seeded shapes: number of contigs, row order, padding, threshold, minimum cluster size, and
    both kernel families (the class's default reference bits, the fast kernels)."""
    rng = np.random.default_rng(4000 + seed)
    _columnar_against_objects(oracle_model, 4100 + seed, int(rng.integers(1, 60)), bool(rng.random() < 0.5), bool(rng.random() < 0.7),
                              float(rng.choice([0.05, 0.3, 0.5, 0.8])), int(rng.integers(1, 4)), min_clusters=0,
                              reference_bits=None if rng.random() < 0.5 else False)


def _columnar_against_objects(oracle_model, seed, n_contigs, shuffle, pad, threshold, n_cds, min_clusters, reference_bits=None):
    import warnings

    from gecco_amd import predict, tables
    from gecco_amd.crf import ClusterCRF
    from gecco_amd.refine import ClusterRefiner

    rng = np.random.default_rng(seed)
    genes_t, feats_t = _random_tables(rng, n_contigs, oracle_model["attrs"][:400], shuffle)
    crf = ClusterCRF.trained(GOLDEN)
    crf.reference_bits = reference_bits
    with warnings.catch_warnings(record=True) as w_col:
        warnings.simplefilter("always")
        g_out, f_out, c_out = predict.predict_tables(genes_t, feats_t, crf, pad=pad, threshold=threshold, n_cds=n_cds)
    # object path on the same data
    by_pid = {g.protein.id: g for g in genes_t.to_genes()}
    for g in feats_t.to_genes():
        by_pid[g.protein.id].protein.domains.extend(g.protein.domains)
    with warnings.catch_warnings(record=True) as w_obj:
        warnings.simplefilter("always")
        annotated = crf.predict_probabilities(list(by_pid.values()), pad=pad)
    assert sorted(str(x.message) for x in w_col) == sorted(str(x.message) for x in w_obj)
    assert list(g_out.protein_id) == [g.protein.id for g in annotated]
    exp_p = np.array([np.nan if g.average_probability is None else g.average_probability for g in annotated])
    np.testing.assert_array_equal(np.isnan(g_out.average_p), np.isnan(exp_p))
    if not np.isnan(exp_p).all():
        assert np.nanmax(np.abs(g_out.average_p - exp_p)) == 0.0  # same kernels, same order: bit-identical
    refiner = ClusterRefiner(threshold=threshold, n_cds=n_cds)
    clusters = []
    import itertools

    for _, group in itertools.groupby(annotated, key=lambda g: g.source.id):
        clusters.extend(refiner.iter_clusters(list(group)))
    exp_t = tables.ClusterTable.from_clusters(clusters)
    assert len(c_out) == len(exp_t) and len(exp_t) >= min_clusters
    for name in ("sequence_id", "cluster_id", "start", "end", "average_p", "max_p", "proteins", "domains"):
        assert list(c_out.columns[name]) == list(exp_t.columns[name]), name
    # every domain row carries its gene's probability
    p_of = {g.protein.id: g.average_probability for g in annotated}
    exp_fp = np.array([np.nan if p_of[pid] is None else p_of[pid] for pid in feats_t.protein_id])
    np.testing.assert_array_equal(f_out.cluster_probability, exp_fp)


def test_columnar_predict_refuses_inconsistent_tables(oracle_model):
    """The reference raises when the tables do not describe the same genes (annotate_genes); so does this path."""
    from gecco_amd import predict, tables
    from gecco_amd.crf import ClusterCRF

    rng = np.random.default_rng(22)
    genes_t, feats_t = _random_tables(rng, 3, oracle_model["attrs"][:50], False)
    crf = ClusterCRF.trained(GOLDEN)
    cols = {k: np.asarray(v).copy() for k, v in feats_t.columns.items()}
    cols["protein_id"][0] = "not_in_the_gene_table"
    with pytest.raises(ValueError, match="missing from the gene table"):
        predict.predict_tables(genes_t, tables.FeatureTable(cols), crf)
    gcols = {k: np.asarray(v).copy() for k, v in genes_t.columns.items()}
    gcols["protein_id"][1] = gcols["protein_id"][0]
    with pytest.raises(ValueError, match="duplicated protein ids"):
        predict.predict_tables(tables.GeneTable(gcols), feats_t, crf)


def test_filter_features_is_filter_domains():
    from gecco_amd import predict, tables

    f = tables.FeatureTable.load(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    assert predict.filter_features(f, None, 1e-9) is f  # the fixture's largest p-value is 3e-10
    g = predict.filter_features(f, None, 1e-20)
    keep = np.asarray(f.pvalue) < 1e-20
    assert 0 < len(g) == int(keep.sum()) < len(f)
    assert list(g.domain) == [d for d, k in zip(f.domain, keep) if k]
    h = predict.filter_features(f, 1e-30, None)
    assert len(h) == int((np.asarray(f.i_evalue) < 1e-30).sum())


def test_columnar_predict_antismash_equals_object_path(oracle_model):
    """`--postproc antismash` (cli/commands/_parser.py:294-301) through the columnar path -- markers found by the native
    packer among ALL domains, criterion evaluated on the device -- against ClusterRefiner(criterion="antismash") on objects."""
    import itertools

    from gecco_amd import predict, refine, tables
    from gecco_amd.crf import ClusterCRF

    rng = np.random.default_rng(23)
    bio = sorted(refine.BIO_PFAMS)
    known = [a for a in oracle_model["attrs"][:60]]
    names = known + bio[:40] + ["PF99999", "TIGR00001"]  # model attributes, marker domains (some unknown to the CRF), strangers
    genes_t, feats_t = _random_tables(rng, 40, names, True)
    crf = ClusterCRF.trained(GOLDEN)
    kw = dict(threshold=0.3, n_cds=2, n_biopfams=2, average_threshold=0.45)
    g_out, f_out, c_out = predict.predict_tables(genes_t, feats_t, crf, criterion="antismash", **kw)
    by_pid = {g.protein.id: g for g in genes_t.to_genes()}
    for g in feats_t.to_genes():
        by_pid[g.protein.id].protein.domains.extend(g.protein.domains)
    annotated = crf.predict_probabilities(list(by_pid.values()))
    refiner = refine.ClusterRefiner(criterion="antismash", **kw)
    clusters = []
    for _, group in itertools.groupby(annotated, key=lambda g: g.source.id):
        clusters.extend(refiner.iter_clusters(list(group)))
    exp_t = tables.ClusterTable.from_clusters(clusters)
    assert len(exp_t) >= 3
    n_gecco = len(predict.predict_tables(genes_t, feats_t, crf, threshold=0.3, n_cds=2)[2])
    assert n_gecco != len(exp_t)  # the criterion matters on this data
    for name in ("sequence_id", "cluster_id", "start", "end", "average_p", "max_p", "proteins", "domains"):
        assert list(c_out.columns[name]) == list(exp_t.columns[name]), name
    with pytest.raises(ValueError, match="Unknown cluster filtering criterion"):
        predict.predict_tables(genes_t, feats_t, crf, criterion="other")
