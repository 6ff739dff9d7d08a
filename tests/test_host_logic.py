"""CPU tests of the host side of the drop-in: model pickle handling, packing, the
``ClusterCRF`` wrapper semantics (sorting, padding, warnings, annotation, cluster weights),
the refiner and the TSV tables.  Where a test needs window scores without a GPU the ORACLE
stands in for the engine (tests may do that; the product never does)."""
import math
import os
import pickle
import shutil
import statistics
import warnings

import itertools

import numpy as np
import pytest

from gecco_amd import crf as crf_mod
from gecco_amd import packing, pickle_model, refine, tables
from gecco_amd.model import Cluster, Domain, Gene, Protein, Source, Strand
from tests.helpers import GOLDEN, read_tsv


@pytest.fixture(scope="module")
def trained():
    return crf_mod.ClusterCRF.trained(GOLDEN)


@pytest.fixture()
def oracle_engine(monkeypatch, oracle_model):
    """Route Model.windowed_marginals to the CPU oracle so that host logic runs without a GPU."""
    from gecco_amd import _native
    from oracle import crf_oracle as orc

    def fake(self, contig_ptr, gene_ptr, attr_id, window, step=1, label=1, pad=True, device=0):
        return orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], contig_ptr, gene_ptr, attr_id,
                                      window, step, label, pad)

    monkeypatch.setattr(_native.Model, "windowed_marginals", fake)

    class FakeSession:  # stands in for the native batch driver (gecco_crf_session_*) on a box without a GPU
        def windowed_marginals(self, contig_ptr, gene_ptr, attr_id, window, step=1, label=1, pad=True, out=None):
            return fake(None, contig_ptr, gene_ptr, attr_id, window, step, label, pad)

        def clusters(self, contig_ptr, gene_ptr, attr_id, annotated, window, step=1, label=1, pad=True, threshold=0.8,
                     n_cds=3, edge_distance=0, trim=True, want_p=False, want_seg_p=True, p_out=None, **_refine_kw):
            p = self.windowed_marginals(contig_ptr, gene_ptr, attr_id, window, step, label, pad)
            seg = orc.segment(p, annotated, contig_ptr, threshold, n_cds, edge_distance, trim, carry_state=False)
            off = np.concatenate([[0], np.cumsum(seg[:, 3] - seg[:, 2])]).astype(np.int64) if len(seg) else np.zeros(1, np.int64)
            seg_p = np.concatenate([p[a:b] for _, _, a, b in seg.tolist()]) if len(seg) and want_seg_p else (np.zeros(0) if want_seg_p else None)
            return seg, seg_p, off, (p if want_p else None)

    from gecco_amd.crf import ClusterCRF

    monkeypatch.setattr(ClusterCRF, "_session", lambda self: FakeSession())


def _gene(contig, pid, start, domains):
    return Gene(Source(contig), start, start + 100, Strand.Coding,
                Protein(pid, None, [Domain(n, s, s + 10, "Pfam", 1e-10, 1e-12) for n, s in domains]))


# ---------------------------------------------------------------- model pickle
def test_trained_attributes(trained):
    assert (trained.feature_type, trained.window_size, trained.window_step, trained.algorithm) == ("protein", 20, 1, "lbfgs")
    assert len(trained.significance) == 11064 and len(trained.significant_features) == 2766
    m = trained.model
    assert m.classes_ == ["0", "1"] and len(m.attributes_) == 2659
    assert (m.c1, m.c2) == (0.4, 0.0)
    sf = m.state_features_
    assert len(sf) == 4211
    # [EXT] %f-rounded like sklearn-crfsuite's parsed dump
    assert sf[("PF13471", "1")] == 3.465119 and sf[("PF00750", "0")] == 0.042199
    assert m.transition_features_ == {("0", "0"): 2.669891, ("0", "1"): -2.599572, ("1", "0"): -2.601921, ("1", "1"): 2.568323}


def test_md5_mismatch(tmp_path):
    shutil.copy(os.path.join(GOLDEN, "model.pkl"), tmp_path / "model.pkl")
    (tmp_path / "model.pkl.md5").write_text("0" * 32)
    with pytest.raises(ValueError, match="MD5 hash of model data does not match signature"):
        crf_mod.ClusterCRF.trained(str(tmp_path))
    # case-insensitive comparison like crf/__init__.py:96
    (tmp_path / "model.pkl.md5").write_text(open(os.path.join(GOLDEN, "model.pkl.md5")).read().upper() + "\n")
    assert crf_mod.ClusterCRF.trained(str(tmp_path)).window_size == 20


def test_unpickler_refuses_foreign_globals(tmp_path):
    import hashlib

    data = pickle.dumps(os.system)
    (tmp_path / "model.pkl").write_bytes(data)
    (tmp_path / "model.pkl.md5").write_text(hashlib.md5(data).hexdigest())
    with pytest.raises(pickle.UnpicklingError):
        crf_mod.ClusterCRF.trained(str(tmp_path))


def test_save_roundtrip_keeps_reference_class_paths(trained, tmp_path):
    trained.save(tmp_path / "out")
    data = (tmp_path / "out" / "model.pkl").read_bytes()
    for path in (b"gecco.crf", b"ClusterCRF", b"sklearn_crfsuite.estimator", b"sklearn_crfsuite._fileresource",
                 b"pycrfsuite._logparser"):
        assert path in data
    assert b"gecco_amd" not in data
    again = crf_mod.ClusterCRF.trained(tmp_path / "out")
    assert again.window_size == 20 and again.model.state_features_ == trained.model.state_features_


def test_constructor_errors():
    with pytest.raises(ValueError, match="invalid feature type: 'gene'"):
        crf_mod.ClusterCRF("gene")
    with pytest.raises(ValueError, match="Window size must be strictly positive"):
        crf_mod.ClusterCRF(window_size=0)
    with pytest.raises(ValueError, match="Window step must be strictly positive and under `window_size`"):
        crf_mod.ClusterCRF(window_size=5, window_step=6)
    c = crf_mod.ClusterCRF(window_size=7, c1=0.1)
    assert c.model is None and c._options == {"algorithm": "lbfgs", "c1": 0.1}
    with pytest.raises(crf_mod.NotFittedError):
        c.predict_probabilities([])


# ---------------------------------------------------------------- packing
def test_feature_extraction_matches_reference_unit_test():
    """tests/test_crf/test_features.py:49-64 of the reference: domain mode -> [{A},{B},{C}],
    protein mode -> [{A,B},{C}]."""
    genes = [_gene("c", "prot1", 0, [("A", 0), ("B", 0)]), _gene("c", "prot2", 1, [("C", 0)])]
    idx = {"A": 0, "B": 1, "C": 2}
    b = packing.pack_contigs([genes], idx, "protein")
    assert b.item_ptr.tolist() == [0, 2] and b.attr_ptr.tolist() == [0, 2, 3] and b.attr_id.tolist() == [0, 1, 2]
    b = packing.pack_contigs([genes], idx, "domain")
    assert b.item_ptr.tolist() == [0, 3] and b.attr_ptr.tolist() == [0, 1, 2, 3] and b.attr_id.tolist() == [0, 1, 2]


def test_packing_collapses_duplicates_and_drops_unknown():
    genes = [_gene("c", "p1", 0, [("A", 0), ("X", 5), ("A", 9)]), _gene("c", "p2", 1, []), _gene("c", "p3", 2, [("X", 0)])]
    b = packing.pack_contigs([genes], {"A": 7}, "protein")
    assert b.attr_ptr.tolist() == [0, 1, 1, 1] and b.attr_id.tolist() == [7]
    b = packing.pack_contigs([genes], {"A": 7}, "domain")  # empty gene -> one empty item
    assert b.item_ptr.tolist() == [0, 5] and b.attr_ptr.tolist() == [0, 1, 1, 2, 2, 2]


def test_columnar_packer_equals_object_packer(trained):
    feats = tables.FeatureTable.load(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    genes_t = tables.GeneTable.load(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    idx = trained.model._attr_index
    cids, order, cptr, gptr, attr, ann = packing.pack_columns(
        feats.sequence_id, feats.protein_id, feats.start, feats.domain, feats.domain_start, idx,
        genes_t.sequence_id, genes_t.protein_id, genes_t.start)
    assert cids == ["BGC0001866.1"] and list(order) == list(genes_t.protein_id) and cptr.tolist() == [0, 23]
    from tests.helpers import golden_csr

    _, cptr2, gptr2, attr2, _, ann2 = golden_csr(idx)
    assert gptr.tolist() == gptr2.tolist() and attr.tolist() == attr2.tolist() and ann.tolist() == ann2.tolist()


def test_vectorised_column_packer_equals_row_by_row_statement():
    """`pack_columns` (hash factorisation + stable sorts) against `pack_columns_py` (the row-by-row
    statement of the reference's ordering rules) on random tables: ties in (contig, start), proteins
    missing from the gene table, repeated gene rows, repeated and unknown domains, empty tables."""
    rng = np.random.default_rng(0)
    for trial in range(200):
        ng, nf = int(rng.integers(0, 40)), int(rng.integers(0, 80))
        nsid, ndom = int(rng.integers(1, 5)), int(rng.integers(1, 9))
        sids = [f"c{int(i):03d}" for i in rng.integers(0, nsid, size=ng)]
        pids = [f"p{i}" for i in range(ng)]
        starts = rng.integers(0, 50, size=ng).tolist()
        if trial % 7 == 0 and ng > 3:
            pids[3] = pids[1]
        fi = rng.integers(0, ng + 3, size=nf)
        args = ([sids[i] if i < ng else "zz" for i in fi], [pids[i] if i < ng else f"x{i}" for i in fi],
                [starts[i] if i < ng else 7 for i in fi], [f"D{int(i)}" for i in rng.integers(0, ndom, size=nf)],
                rng.integers(0, 6, size=nf).tolist(), {f"D{i}": i * 2 for i in range(0, ndom, 2)})
        if trial % 3:
            args += (sids, pids, starts)
        a, b = packing.pack_columns_py(*args), packing.pack_columns(*args)
        assert a[0] == b.contig_ids and list(a[1]) == list(b.order), trial
        assert all(np.array_equal(x, y) for x, y in zip(a[2:], list(b)[2:])), trial
        # the extra row bookkeeping: rows of gene k, by domain start, stable
        for k in range(len(b.order)):
            rows = b.row_order[b.row_ptr[k]:b.row_ptr[k + 1]]
            exp = sorted([r for r in range(nf) if args[1][r] == b.order[k]], key=lambda r: args[4][r])
            assert rows.tolist() == exp, trial


# ---------------------------------------------------------------- predict_probabilities wrapper
def _golden_genes():
    feats = tables.FeatureTable.load(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    genes_t = tables.GeneTable.load(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    annotated = {g.protein.id: g for g in feats.to_genes()}
    out = []
    for g in genes_t.to_genes():  # genes.tsv also lists the genes without any domain
        a = annotated.get(g.protein.id)
        out.append(Gene(g.source, g.start, g.end, g.strand, a.protein if a else g.protein))
    return out


def test_predict_probabilities_reproduces_golden_tables(trained, oracle_engine):
    genes = _golden_genes()
    rng = np.random.default_rng(0)
    shuffled = [genes[i] for i in rng.permutation(len(genes))]
    calls = []
    out = trained.predict_probabilities(shuffled, progress=lambda i, t: calls.append((i, t)))
    assert [g.protein.id for g in out] == [r["protein_id"] for r in read_tsv(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))]
    exp = [float(r["average_p"]) for r in read_tsv(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))]
    assert max(abs(g.average_probability - e) for g, e in zip(out, exp)) <= 1e-15
    assert calls[0] == (0, 4) and calls[-1] == (4, 4)  # 23 genes, W=20 -> 4 windows
    # domains carry the gene's probability and the %f-rounded state weight of (name,'1')
    sf = trained.model.state_features_
    for g in out:
        for d in g.protein.domains:
            assert d.probability == g.average_probability
            assert d.cluster_weight == sf.get((d.name, "1"))
    # table writers reproduce the reference's files (numerically; text equality up to last-digit rounding)
    import io

    buf = io.StringIO()
    tables.GeneTable.from_genes(out).dump(buf)
    got = [l.split("\t") for l in buf.getvalue().strip().split("\n")]
    ref = [l.split("\t") for l in open(os.path.join(GOLDEN, "BGC0001866.genes.tsv")).read().strip().split("\n")]
    assert got[0] == ref[0] and len(got) == len(ref)
    for a, b in zip(got[1:], ref[1:]):
        assert a[:5] == b[:5] and abs(float(a[5]) - float(b[5])) <= 1e-15 and abs(float(a[6]) - float(b[6])) <= 1e-15
    buf = io.StringIO()
    tables.FeatureTable.from_genes(out).dump(buf)
    got = [l.split("\t") for l in buf.getvalue().strip().split("\n")]
    ref = [l.split("\t") for l in open(os.path.join(GOLDEN, "BGC0001866.features.tsv")).read().strip().split("\n")]
    assert got[0] == ref[0] and len(got) == len(ref) == 38
    for a, b in zip(got[1:], ref[1:]):
        assert a[:7] == b[:7] and a[9:11] == b[9:11]
        assert float(a[7]) == float(b[7]) and float(a[8]) == float(b[8]) and abs(float(a[11]) - float(b[11])) <= 1e-15

    # cluster calling on top: exactly one cluster, same row as clusters.tsv (CRF/refiner columns)
    clusters = list(refine.ClusterRefiner(threshold=0.8, n_cds=3, edge_distance=0, trim=True).iter_clusters(out))
    row = read_tsv(os.path.join(GOLDEN, "BGC0001866.clusters.tsv"))[0]
    assert len(clusters) == 1
    c = clusters[0]
    assert (c.source.id, c.id, c.start, c.end) == (row["sequence_id"], row["cluster_id"], int(row["start"]), int(row["end"]))
    assert abs(c.average_probability - float(row["average_p"])) <= 2e-16
    assert abs(c.maximum_probability - float(row["max_p"])) <= 1e-15
    assert {g.protein.id for g in c.genes} == set(row["proteins"].split(";"))
    assert sorted({d.name for g in c.genes for d in g.protein.domains}) == row["domains"].split(";")


def test_padding_and_skipping_semantics(trained, oracle_engine):
    short = [_gene("tiny", f"tiny_{i}", 10 * i, [("PF00109", 1)] if i % 2 else []) for i in range(7)]
    long_ = [_gene("big", f"big_{i}", 10 * i, [("PF00106", 1), ("PF08659", 30)] if i % 3 == 0 else []) for i in range(25)]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = trained.predict_probabilities(short + long_, pad=True)
    assert [str(x.message) for x in w] == [
        "Contig 'tiny' does not contain enough proteins (7) for sliding window of size 20, padding with 13 proteins"
    ]
    assert [g.source.id for g in out] == ["big"] * 25 + ["tiny"] * 7  # sorted by contig id
    assert all(g.average_probability is not None for g in out)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out2 = trained.predict_probabilities(short + long_, pad=False)
    assert [str(x.message) for x in w] == [
        "Contig 'tiny' does not contain enough proteins (7) for sliding window of size 20"
    ]
    tiny = [g for g in out2 if g.source.id == "tiny"]
    assert all(g.average_probability is None for g in tiny)  # passed through without prediction
    assert [g.average_probability for g in out2[:25]] == [g.average_probability for g in out[:25]]
    # singular unit in the warning (window - len == 1)
    c19 = [_gene("n19", f"n19_{i}", i, []) for i in range(19)]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        trained.predict_probabilities(c19)
    assert str(w[0].message).endswith("padding with 1 protein")


def test_domains_are_sorted_in_place_like_the_reference(trained, oracle_engine):
    g = _gene("c", "p", 0, [("PF00106", 50), ("PF08659", 5)])
    genes = [g] + [_gene("c", f"q{i}", 10 + i, []) for i in range(20)]
    trained.predict_probabilities(genes)
    assert [d.start for d in g.protein.domains] == [5, 50]  # caller's list mutated (crf/__init__.py:200-201)


# ---------------------------------------------------------------- refiner
def _pgene(contig, i, p, annotated=True):
    return Gene(Source(contig), 100 * i, 100 * i + 90, Strand.Coding,
                Protein(f"{contig}_{i}", None, [Domain("PF00109", 1, 9, "Pfam", 1e-9, 1e-9)] if annotated else []),
                _probability=p)


def test_refiner_matches_oracle_segment():
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(11)
    genes, p_all, ann_all, cptr = [], [], [], [0]
    for c in range(30):
        n = int(rng.integers(1, 60))
        base = rng.random() < 0.5
        p = np.clip(rng.normal(0.85 if base else 0.3, 0.3, size=n), 0, 1)
        p[rng.random(n) < 0.05] = np.nan
        ann = rng.random(n) < 0.7
        for i in range(n):
            genes.append(_pgene(f"ctg{c:03d}", i, None if np.isnan(p[i]) else float(p[i]), bool(ann[i])))
        p_all += p.tolist()
        ann_all += ann.tolist()
        cptr.append(cptr[-1] + n)
    for n_cds, edge, trim in [(3, 0, True), (1, 0, False), (2, 2, True), (5, 1, True)]:
        # one iter_clusters call over every contig: one grouper, its state carries across contigs
        seg = orc.segment(np.array(p_all), np.array(ann_all, dtype=np.uint8), np.array(cptr), 0.8, n_cds, edge, trim,
                          carry_state=True)
        rng.shuffle(genes)
        refiner = refine.ClusterRefiner(threshold=0.8, n_cds=n_cds, edge_distance=edge, trim=trim, cluster_type=Cluster)
        got = list(refiner.iter_clusters(genes))
        exp = [(f"ctg{c:03d}_cluster_{k}", [f"ctg{c:03d}_{g - cptr[c]}" for g in range(a, b)]) for c, k, a, b in seg.tolist()]
        assert [(c.id, [g.id for g in c.genes]) for c in got] == exp
        # one call per contig, what the CLI does (cli/commands/_common.py:621-623): a fresh grouper each time
        seg0 = orc.segment(np.array(p_all), np.array(ann_all, dtype=np.uint8), np.array(cptr), 0.8, n_cds, edge, trim,
                           carry_state=False)
        got0 = []
        for _, group in itertools.groupby(sorted(genes, key=lambda g: g.source.id), key=lambda g: g.source.id):
            got0.extend(refiner.iter_clusters(list(group)))
        exp0 = [(f"ctg{c:03d}_cluster_{k}", [f"ctg{c:03d}_{g - cptr[c]}" for g in range(a, b)]) for c, k, a, b in seg0.tolist()]
        assert [(c.id, [g.id for g in c.genes]) for c in got0] == exp0


def test_refiner_defaults_and_antismash():
    r = refine.ClusterRefiner()
    assert (r.threshold, r.criterion, r.n_cds, r.n_biopfams, r.average_threshold, r.edge_distance, r.trim) == \
        (0.8, "gecco", 5, 5, 0.6, 0, True)
    assert len(refine.BIO_PFAMS) == 130 and "PF00109" in refine.BIO_PFAMS
    names = ["PF00109", "PF02801", "PF08659", "PF00378", "PF08541", "PF00550"]
    genes = [Gene(Source("c"), 10 * i, 10 * i + 5, Strand.Coding,
                  Protein(f"g{i}", None, [Domain(names[i], 1, 2, "Pfam", 0, 0)]), _probability=0.9) for i in range(6)]
    assert len(list(refine.ClusterRefiner(criterion="antismash", n_cds=5, n_biopfams=5, cluster_type=Cluster).iter_clusters(genes))) == 1
    assert len(list(refine.ClusterRefiner(criterion="antismash", n_cds=5, n_biopfams=6, cluster_type=Cluster).iter_clusters(genes))) == 0
    with pytest.raises(ValueError, match="Unknown cluster filtering criterion"):
        list(refine.ClusterRefiner(criterion="x").iter_clusters(genes))


def test_oracle_antismash_segmenter_equals_object_refiner():
    """oracle_segment_antismash (the checker of the device path) against ClusterRefiner(criterion="antismash") on
    objects, the mirror of refine.py:118-200."""
    import itertools

    from oracle import crf_oracle as orc

    rng = np.random.default_rng(8)
    markers = sorted(refine.BIO_PFAMS)
    genes, p_all, ann, mptr, mid, cptr = [], [], [], [0], [], [0]
    for c in range(25):
        n = int(rng.integers(1, 90))
        base = 0.9 if rng.random() < 0.5 else 0.4
        for g in range(n):
            names = [str(rng.choice(markers[:12])) if rng.random() < 0.5 else f"PF9{int(rng.integers(0, 30)):04d}"
                     for _ in range(int(rng.integers(0, 4)))]
            p = float(np.clip(rng.normal(base, 0.25), 0, 1))
            doms = [Domain(nm, i, i + 1, "Pfam", 0.0, 0.0, p) for i, nm in enumerate(names)]
            genes.append(Gene(Source(f"c{c:02d}"), 10 * g, 10 * g + 9, Strand.Coding, Protein(f"c{c:02d}_{g}", None, doms), _probability=p))
            p_all.append(p)
            ann.append(1 if doms else 0)
            mid.extend(sorted({markers.index(nm) for nm in names if nm in refine.BIO_PFAMS}))
            mptr.append(len(mid))
        cptr.append(len(p_all))
    total = 0
    for kw in (dict(n_cds=5, n_biopfams=5, average_threshold=0.6), dict(n_cds=2, n_biopfams=2, average_threshold=0.85),
               dict(n_cds=3, n_biopfams=1, average_threshold=0.5, trim=False), dict(n_cds=1, n_biopfams=0, average_threshold=0.0)):
        ref = refine.ClusterRefiner(criterion="antismash", threshold=0.8, cluster_type=Cluster, **kw)
        exp = []
        for ci, (_, group) in enumerate(itertools.groupby(genes, key=lambda g: g.source.id)):
            for cl in ref.iter_clusters(list(group)):
                first = next(i for i in range(cptr[ci], cptr[ci + 1]) if genes[i] is cl.genes[0])
                exp.append([ci, int(cl.id.rsplit("_", 1)[1]), first, first + len(cl.genes)])
        got = orc.segment_antismash(p_all, ann, cptr, mptr, mid, 0.8, kw["n_cds"], kw["n_biopfams"], kw["average_threshold"],
                                    kw.get("trim", True))
        assert got.tolist() == exp
        total += len(exp)
    assert total > 10


def test_cluster_average_is_exactly_rounded():
    ps = [float(r["average_p"]) for r in read_tsv(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))]
    c = Cluster("x", [_pgene("c", i, p) for i, p in enumerate(ps)])
    assert c.average_probability == statistics.mean(ps) == 0.9958958770931705


# ---------------------------------------------------------------- tables
def test_tables_roundtrip(tmp_path):
    for cls, name in ((tables.FeatureTable, "features"), (tables.GeneTable, "genes")):
        src = os.path.join(GOLDEN, f"BGC0001866.{name}.tsv")
        t = cls.load(src)
        t.dump(str(tmp_path / f"{name}.tsv"))
        assert open(src).read() == open(tmp_path / f"{name}.tsv").read()
    g = tables.GeneTable.from_genes([_pgene("c", 0, None), _pgene("c", 1, None)])
    import io

    buf = io.StringIO()
    g.dump(buf)  # all-NaN probability columns are dropped (gecco/_base.py:138-146)
    assert buf.getvalue().split("\n")[0] == "sequence_id\tprotein_id\tstart\tend\tstrand"


def test_tables_empty_cells(tmp_path):
    """An empty cell in an integer column is the row parser's `int("")` error in every loader (never INT64_MIN); a
    None in a numeric column of a bulk table goes to the row-by-row writer instead of failing in the native one."""
    lines = open(os.path.join(GOLDEN, "BGC0001866.genes.tsv")).read().split("\n")
    header = lines[0].split("\t")
    row = lines[1].split("\t")
    row[header.index("start")] = ""
    bad = tmp_path / "bad.tsv"
    bad.write_text("\n".join([lines[0], "\t".join(row)] + lines[2:]))
    with pytest.raises(ValueError, match="invalid literal for int"):
        tables.GeneTable.load(str(bad))
    t = tables.GeneTable.load(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    n = len(t)
    big = tables.GeneTable({k: list(v.to_objects() if isinstance(v, tables.StringColumn) else v) * (70 // n + 1)
                            for k, v in t.columns.items()})
    big.columns["average_p"][3] = None
    import io

    buf = io.StringIO()
    big.dump(buf)
    assert len(buf.getvalue().split("\n")) == len(big) + 2


def test_domain_feature_mode(oracle_engine, trained, monkeypatch):
    """feature_type='domain': one item per domain (one empty item for a gene without domains);
    every domain gets its own probability, genes without domains their item's
    (features.py:38-48,99-120)."""
    import copy

    crf = copy.copy(trained)
    crf.feature_type = "domain"
    genes = []
    for i in range(30):
        doms = [("PF00109", 1), ("PF02801", 40)] if i % 4 == 0 else ([("PF00106", 5)] if i % 4 == 1 else [])
        genes.append(_gene("ctg", f"g{i:02d}", 10 * i, doms))
    out = crf.predict_probabilities(genes)
    n_items = sum(max(1, len(g.protein.domains)) for g in genes)
    assert n_items == 8 * 2 + 8 * 1 + 14 * 1 + 0  # sanity on the construction
    for g in out:
        if g.protein.domains:
            assert g._probability is None and all(d.probability is not None for d in g.protein.domains)
        else:
            assert g._probability is not None
    two = [g for g in out if len(g.protein.domains) == 2]
    assert any(g.protein.domains[0].probability != g.protein.domains[1].probability for g in two)


@pytest.mark.parametrize("threads,grain", [(1, 16384), (5, 3), (16, 1)])
def test_native_column_packer_equals_row_by_row_statement(monkeypatch, threads, grain):
    """`gecco_crf_pack_columns` (csrc/crf_tables.cpp: Arrow-layout strings, interning with adjacency shortcuts,
    order checks + radix sorts) against `pack_columns_py` on the same random tables as above, plus tables that
    are already in order (the fast paths) and coordinates that do not fit the radix keys."""
    from gecco_amd import _native as nat

    # the per-row passes run on several host threads for large tables: cut these tiny ones the same way
    monkeypatch.setenv("GECCO_CRF_HOST_THREADS", str(threads))
    monkeypatch.setenv("GECCO_CRF_HOST_GRAIN", str(grain))
    A = 12
    model = nat.Model.from_tables(np.zeros((A, 2)), np.zeros((2, 2)))  # attribute names "a0" .. "a11"
    idx = {f"a{i}": i for i in range(A)}
    rng = np.random.default_rng(1)
    for trial in range(300):
        ng, nf = int(rng.integers(0, 40)), int(rng.integers(0, 80))
        nsid, ndom = int(rng.integers(1, 5)), int(rng.integers(1, 16))
        sids = [f"c{int(i):03d}" for i in rng.integers(0, nsid, size=ng)]
        pids = [f"p{i}" for i in range(ng)]
        big = 1 << 45 if trial % 11 == 0 else 50
        starts = rng.integers(-3 if trial % 5 == 0 else 0, big, size=ng).tolist()
        if trial % 4 == 0:  # a table GECCO wrote: genes in order, feature rows following them
            order = sorted(range(ng), key=lambda i: (sids[i], starts[i]))
            sids, starts = [sids[i] for i in order], [starts[i] for i in order]
            fi = np.sort(rng.integers(0, max(ng, 1), size=nf)) if ng else rng.integers(0, 3, size=nf)
        else:
            fi = rng.integers(0, ng + 3, size=nf)
        if trial % 7 == 0 and ng > 3:
            pids[3] = pids[1]
        f_sid = [sids[i] if i < ng else "zz" for i in fi]
        f_pid = [pids[i] if i < ng else f"x{i}" for i in fi]
        f_start = [starts[i] if i < ng else 7 for i in fi]
        f_dom = [f"a{int(i)}" if i < A else f"unknown{int(i)}" for i in rng.integers(0, ndom, size=nf)]
        f_ds = rng.integers(0, 6 if trial % 13 else (1 << 40), size=nf).tolist()
        args = (f_sid, f_pid, f_start, f_dom, f_ds, idx)
        S = tables.StringColumn.from_sequence
        gargs = (None, None, None)
        if trial % 3:
            args += (sids, pids, starts)
            gargs = (S(sids), S(pids), np.array(starts, dtype=np.int64))
        a = packing.pack_columns_py(*args)
        # every other table also asks for the genes' marker domains (antismash criterion): names the model knows
        # ("a3") and names it does not ("unknown13")
        markers = ["a3", "unknown13", "a0", "nowhere", "unknown14"] if trial % 2 else None
        b = nat.PackedTables(model, S(f_sid), S(f_pid), np.array(f_start, dtype=np.int64), S(f_dom),
                             np.array(f_ds, dtype=np.int64), *gargs, markers=S(markers) if markers else None)
        n = len(a[1])
        assert b.n_genes == n and b.n_contigs == len(a[0]), trial
        all_pid = (pids if trial % 3 else [])
        names = [all_pid[r] if r >= 0 else f_pid[-1 - r] for r in b.gene_row.tolist()]
        assert names == list(a[1]), trial
        assert b.contig_ptr.tolist() == a[2].tolist() and b.gene_ptr.tolist() == a[3].tolist(), trial
        assert b.attr_id.tolist() == a[4].tolist() and b.annotated.tolist() == a[5].tolist(), trial
        for k in range(n):
            rows = b.row_order[b.row_ptr[k]:b.row_ptr[k + 1]]
            exp = sorted([r for r in range(nf) if f_pid[r] == names[k]], key=lambda r: f_ds[r])
            assert rows.tolist() == exp, trial
            assert all(b.row_gene[r] == k for r in exp), trial
            if markers:  # distinct marker domains among ALL rows of the gene, in row order
                exp_m = list(dict.fromkeys(markers.index(f_dom[r]) for r in exp if f_dom[r] in markers))
                assert b.marker_id[b.marker_ptr[k]:b.marker_ptr[k + 1]].tolist() == exp_m, trial
        assert (b.marker_ptr is None) == (markers is None)
        if trial % 3:
            assert b.n_duplicate_gene_ids == (1 if (trial % 7 == 0 and ng > 3) else 0), trial
            assert b.n_unlisted_proteins == (len({p for p in f_pid if p not in set(pids)}) if ng else 0), trial


def test_exact_mean_is_statistics_mean():
    """average_p of a cluster is `statistics.mean` (gecco/model.py:442-447): exact sum, ONE rounding."""
    import statistics

    from gecco_amd import _native as nat

    rng = np.random.default_rng(2)
    for trial in range(400):
        k = int(rng.integers(1, 60))
        kind = trial % 5
        if kind == 0:
            v = rng.random(k)
        elif kind == 1:
            v = 1.0 - rng.random(k) * 1e-9         # saturated probabilities: the interesting case
        elif kind == 2:
            v = rng.random(k) * 10.0 ** rng.integers(-320, 10, size=k)   # subnormals, huge spread
        elif kind == 3:
            v = np.full(k, 0.1)
        else:
            v = np.ldexp(rng.integers(1, 1 << 53, size=k).astype(np.float64), int(rng.integers(-1074 - 30, -1000)))
        assert nat.exact_mean(v) == statistics.mean(v.tolist()), (trial, v)
    assert np.isnan(nat.exact_mean([float("nan")])) and nat.exact_mean([float("nan"), 0.25, 0.5]) == 0.375
    assert nat.exact_mean([5e-324, 5e-324, 0.0]) == statistics.mean([5e-324, 5e-324, 0.0])


@pytest.mark.parametrize("threads", [1, 3])
def test_native_cluster_rows_match_python_assembly(monkeypatch, threads):
    import statistics

    from gecco_amd import _native as nat

    monkeypatch.setenv("GECCO_CRF_HOST_THREADS", str(threads))
    monkeypatch.setenv("GECCO_CRF_HOST_GRAIN", "1")

    model = nat.Model.from_tables(np.zeros((4, 2)), np.zeros((2, 2)))
    S = tables.StringColumn.from_sequence
    rng = np.random.default_rng(3)
    ng = 60
    sids = ["ctgB"] * 30 + ["ctgA"] * 30
    pids = [f"g{rng.integers(0, 1000):03d}_{i}" for i in range(ng)]
    starts = (np.arange(ng) % 30 * 100).astype(np.int64)
    ends = starts + rng.integers(50, 500, size=ng)
    fi = np.sort(rng.integers(0, ng, size=150))
    f_dom = [f"PF{rng.integers(0, 30):05d}" for _ in fi]
    pk = nat.PackedTables(model, S([sids[i] for i in fi]), S([pids[i] for i in fi]), starts[fi], S(f_dom),
                          rng.integers(0, 300, size=len(fi)).astype(np.int64), S(sids), S(pids), starts)
    order = [pids[r] for r in pk.gene_row.tolist()]
    p = rng.random(ng)
    seg = np.array([[0, 1, 2, 9], [0, 3, 12, 13], [1, 1, 30, 55], [1, 2, 58, 58]], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(seg[:, 3] - seg[:, 2])]).astype(np.int64)
    seg_p = np.concatenate([p[a:b] for _, _, a, b in seg.tolist()])
    cr = pk.cluster_rows(seg, seg_p, off, ends, ends[fi])
    col = lambda name: tables.StringColumn(*cr[name]).tolist()
    for k, (c, number, a, b) in enumerate(seg.tolist()):
        rows = pk.gene_row[a:b]
        cid = ["ctgA", "ctgB"][c]
        assert col("sequence_id")[k] == (cid if b > a else "") and col("cluster_id")[k] == (cid if b > a else "") + f"_cluster_{number}"
        assert col("proteins")[k] == ";".join(sorted(order[a:b]))
        drows = pk.row_order[pk.row_ptr[a]:pk.row_ptr[b]]
        assert col("domains")[k] == ";".join(sorted(f_dom[r] for r in drows))
        if b > a:
            assert cr["start"][k] == starts[rows].min() and cr["end"][k] == ends[rows].max()
            assert cr["average_p"][k] == statistics.mean(p[a:b].tolist()) and cr["max_p"][k] == p[a:b].max()
        else:
            assert np.isnan(cr["average_p"][k]) and np.isnan(cr["max_p"][k])


def test_native_tsv_writer_is_the_row_by_row_writer(monkeypatch):
    """`gecco_crf_tsv_format` (bulk tables) must produce the bytes of the reference-faithful row-by-row writer:
    floats with repr() digits in repr()'s layout, NaN as an empty field (gecco/_base.py:133-152)."""
    import io

    from gecco_amd import _native as nat

    rng = np.random.default_rng(4)
    vals = np.concatenate([rng.random(3000), rng.random(3000) * 10.0 ** rng.integers(-330, 300, size=3000), -rng.random(50),
                           [0.0, -0.0, 1e16, 1e15, 123456789012345680.0, 1e-4, 1e-5, 0.0001234, 5e-324,
                            1.7976931348623157e308, np.nan, np.inf, -np.inf, 100.0, 1e22, 0.1, 2.0 ** 53, 9999999999999998.0]])
    got = nat.tsv_format("x\n", [vals]).decode().split("\n")[1:-1]
    assert got == ["" if v != v else repr(float(v)) for v in vals]
    # a whole table, several host threads: same bytes as the small-table (row by row) path
    monkeypatch.setenv("GECCO_CRF_HOST_THREADS", "4")
    monkeypatch.setenv("GECCO_CRF_HOST_GRAIN", "8")
    n = 300
    cols = {"sequence_id": np.array([f"c{i // 50}" for i in range(n)], dtype=object),
            "protein_id": np.array([f"g{i}_é" for i in range(n)], dtype=object), "start": np.arange(n) * 10 - 5,
            "end": np.arange(n) * 10 + 9, "strand": np.full(n, "+", dtype=object),
            "average_p": np.where(rng.random(n) < 0.2, np.nan, rng.random(n)), "max_p": rng.random(n) * 1e-7}
    t = tables.GeneTable(dict(cols))
    bulk = io.StringIO()
    t.dump(bulk)
    names = t._dump_columns()
    rows = ["\t".join(names)] + ["\t".join(tables._fmt(np.asarray(t.columns[k], dtype=object)[i] if not isinstance(t.columns[k], np.ndarray)
                                                          or t.columns[k].dtype == object else t.columns[k][i]) for k in names)
                                 for i in range(n)]
    assert bulk.getvalue() == "\n".join(rows) + "\n"
    back = tables.GeneTable.load(io.BytesIO(bulk.getvalue().encode()))
    assert list(back.protein_id) == list(cols["protein_id"]) and np.array_equal(np.asarray(back.start), cols["start"])
    np.testing.assert_array_equal(np.asarray(back.average_p), cols["average_p"])


def test_gecco_hip_entry_point(monkeypatch, capsys):
    """`gecco-hip` = gecco.cli.main(crf_type=gecco_amd.crf.ClusterCRF) (cli/commands/__init__.py:127-137); without GECCO
    installed it says what it needs instead of a traceback."""
    import sys
    import types

    from gecco_amd import cli
    from gecco_amd.crf import ClusterCRF

    monkeypatch.setitem(sys.modules, "gecco", None)  # import gecco -> ImportError
    assert cli.main(["run", "--help"]) == 2
    assert "needs GECCO itself" in capsys.readouterr().err
    seen = {}
    fake = types.ModuleType("gecco")
    fake.cli = types.ModuleType("gecco.cli")

    def fake_main(argv=None, console=None, *, crf_type=None, **kw):
        seen.update(argv=argv, crf_type=crf_type)
        return 0

    fake.cli.main = fake_main
    monkeypatch.setitem(sys.modules, "gecco", fake)
    monkeypatch.setitem(sys.modules, "gecco.cli", fake.cli)
    assert cli.main(["predict", "-o", "x"]) == 0
    assert seen == {"argv": ["predict", "-o", "x"], "crf_type": ClusterCRF}


def test_duck_typed_models_go_through_their_own_with_methods(trained, oracle_engine):
    """Genes that are not GECCO-style dataclasses (slots, other field names) are annotated through the with_* methods
    the reference itself calls (model.py:364-375, crf/__init__.py:261-269); same probabilities and weights as the
    dataclass fast path."""
    class D:
        __slots__ = ("name", "start", "end", "probability", "cluster_weight")

        def __init__(self, name, start, end, probability=None, cluster_weight=None):
            self.name, self.start, self.end, self.probability, self.cluster_weight = name, start, end, probability, cluster_weight

        def with_probability(self, p):
            return D(self.name, self.start, self.end, p, self.cluster_weight)

        def with_cluster_weight(self, w):
            return D(self.name, self.start, self.end, self.probability, w)

    class P:
        def __init__(self, id, domains):
            self.id, self.domains = id, list(domains)

        def with_domains(self, domains):
            return P(self.id, domains)

    class G:
        def __init__(self, source, start, end, protein, p=None):
            self.source, self.start, self.end, self.protein, self.p = source, start, end, protein, p

        def with_probability(self, p):
            return G(self.source, self.start, self.end, self.protein.with_domains([d.with_probability(p) for d in self.protein.domains]), p)

        def with_protein(self, protein):
            return G(self.source, self.start, self.end, protein, self.p)

    ref = _golden_genes()
    duck = [G(g.source, g.start, g.end, P(g.protein.id, [D(d.name, d.start, d.end) for d in g.protein.domains])) for g in ref]
    a = trained.predict_probabilities(ref)
    b = trained.predict_probabilities(duck)
    assert [g.protein.id for g in a] == [g.protein.id for g in b]
    assert [g.average_probability for g in a] == [g.p for g in b]
    for ga, gb in zip(a, b):
        assert [(d.name, d.probability, d.cluster_weight) for d in ga.protein.domains] == \
            [(d.name, d.probability, d.cluster_weight) for d in gb.protein.domains]
    assert all(type(g) is G and type(g.protein) is P for g in b)


def test_objpath_extension_equals_python_statements(monkeypatch):
    """csrc/objpath.c (CPython C API) against the Python statements of the same two loops: repeated domain names
    collapse, unknown names drop, more than sixteen domains per gene, genes without domains; the clones are new objects
    that compare equal to what the Python loop builds; an object of another class makes the fast path give up."""
    import numpy as np

    from gecco_amd import _objpath_loader, crf as crf_mod, packing
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand

    native = _objpath_loader.module()
    assert native is not None, "gecco_amd/csrc/objpath.c did not build"
    rng = np.random.default_rng(5)
    names = [f"PF{i:05d}" for i in range(60)]
    attr_index = {n: i for i, n in enumerate(names[:40])}  # the last 20 names are unknown to the model
    contigs = []
    for c in range(7):
        src = Source(f"c{c}")
        genes = []
        for i in range(int(rng.integers(0, 40))):
            k = int(rng.choice([0, 1, 2, 3, 5, 20]))
            doms = [Domain(names[int(a)], 3 * j, 3 * j + 2, "Pfam", 1e-9, 1e-11, qualifiers={"x": [str(j)]})
                    for j, a in enumerate(rng.integers(0, 60, size=k))]
            genes.append(Gene(src, 10 * i, 10 * i + 9, Strand.Coding, Protein(f"c{c}_{i}", None, doms), qualifiers={"g": ["q"]}))
        contigs.append(genes)
    got = packing.pack_contigs(contigs, attr_index, "protein")
    w1 = {n: float(i) for i, n in enumerate(names[:30])}
    probs = [[float(x) for x in rng.random(len(c))] for c in contigs]
    fast = [crf_mod._annotate_all(c, p, w1) for c, p in zip(contigs, probs)]
    monkeypatch.setenv("GECCO_AMD_NO_OBJPATH", "1")
    monkeypatch.setattr(_objpath_loader, "_tried", False)
    monkeypatch.setattr(_objpath_loader, "_mod", None)
    exp = packing.pack_contigs(contigs, attr_index, "protein")
    slow = [crf_mod._annotate_all(c, p, w1) for c, p in zip(contigs, probs)]
    for a, b in ((got.item_ptr, exp.item_ptr), (got.attr_ptr, exp.attr_ptr), (got.attr_id, exp.attr_id)):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    assert fast == slow
    for c, f in zip(contigs, fast):
        for g, ng in zip(c, f):
            assert ng is not g and ng.protein is not g.protein and ng.qualifiers is not g.qualifiers and ng.qualifiers == g.qualifiers
            for d, nd in zip(g.protein.domains, ng.protein.domains):
                assert nd is not d and nd.qualifiers is not d.qualifiers and nd.cluster_weight == w1.get(d.name)
    # an object of another class: the C loop returns None like the Python one
    class Other:
        pass

    o = Other()
    o.__dict__.update(contigs[0][0].__dict__) if contigs[0] else None
    if contigs[1]:
        assert native.annotate_all([contigs[1][0], o], [0.1, 0.2], w1, Gene, Protein, Domain) is None


def test_committed_profiles_parse():
    """Every profiles/*.json is one JSON document (launcher log lines in front of a bench line made two of them unreadable)."""
    import glob
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "*.json")))
    assert files
    for f in files:
        with open(f) as fh:
            json.load(fh)


def test_objpath_sort_group_equals_the_python_statements():
    """csrc/objpath.c sort_group against gecco/crf/__init__.py:199-206 as Python states it: an input already in
    (source.id, start) order comes back as it is, grouped, with unsorted domain lists sorted in place; any other input
    returns None (the caller then runs the Python statements)."""
    import itertools
    import operator

    from gecco_amd import _objpath_loader
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand

    native = _objpath_loader.module()
    assert native is not None, "gecco_amd/csrc/objpath.c did not build"

    def make(order):
        srcs = {k: Source(k) for k in "abc"}
        out = []
        for sid, start, dstarts in order:
            doms = [Domain(f"PF{d:05d}", d, d + 5, "Pfam", 1e-10, 1e-12) for d in dstarts]
            out.append(Gene(srcs[sid], start, start + 90, Strand.Coding, Protein(f"{sid}_{start}", None, doms)))
        return out

    ordered = [("a", 1, [3, 1, 2]), ("a", 1, []), ("a", 7, [5]), ("b", 2, [9, 9, 4]), ("c", 0, [1, 2])]
    genes = make(ordered)
    ref = make(ordered)
    ref_sorted = sorted(ref, key=operator.attrgetter("source.id", "start"))
    for g in ref_sorted:
        g.protein.domains.sort(key=operator.attrgetter("start"))
    ref_contigs = [list(g) for _, g in itertools.groupby(ref_sorted, key=operator.attrgetter("source.id"))]
    got = native.sort_group(genes, operator.attrgetter("start"))
    assert got is not None
    out_genes, contigs = got
    assert [id(g) for g in out_genes] == [id(g) for g in genes]  # (stable: the input order)
    assert [[g.protein.id for g in c] for c in contigs] == [[g.protein.id for g in c] for c in ref_contigs]
    assert [[d.start for d in g.protein.domains] for g in out_genes] == [[d.start for d in g.protein.domains] for g in ref_sorted]
    # unsorted inputs: contig ids out of order, starts out of order inside a contig, a generator
    assert native.sort_group(make([("b", 1, []), ("a", 2, [])]), operator.attrgetter("start")) is None
    assert native.sort_group(make([("a", 5, []), ("a", 2, [])]), operator.attrgetter("start")) is None
    got = native.sort_group(iter(make(ordered)), operator.attrgetter("start"))
    assert got is not None and len(got[0]) == 5 and len(got[1]) == 3
    assert native.sort_group([], operator.attrgetter("start")) == ([], [])


def test_objpath_fused_pass_and_buffer_annotation_equal_the_separate_steps():
    """sort_group(genes, key, attr_index) -- the pass that also packs -- returns pack_protein's arrays of the grouped contigs
    (after the in-place domain sorts: feature order follows the SORTED domain list, duplicates and unknown names dropped; equal
    source ids on different Source objects are one contig; objects with properties and slots go through getattr), and
    annotate_all on a float64 buffer builds the objects annotate_all builds from a list of floats."""
    import operator

    import numpy as np

    from gecco_amd import _objpath_loader, packing
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand

    native = _objpath_loader.module()
    assert native is not None
    rng = np.random.default_rng(5)
    names = [f"PF{k:05d}" for k in range(40)]
    index = {nm: i for i, nm in enumerate(names[:30])}  # (the last ten names: unknown to the model)
    genes = []
    for c in range(6):
        for i in range(int(rng.integers(1, 30))):
            src = Source(f"contig_{c}")  # a fresh Source per gene: equal ids, different objects
            k = int(rng.integers(0, 5))
            doms = [Domain(names[int(a)], int(s), int(s) + 5, "Pfam", 1e-10, 1e-12)
                    for a, s in zip(rng.integers(0, 40, size=k), rng.integers(0, 50, size=k))]
            genes.append(Gene(src, 10 * i, 10 * i + 9, Strand.Coding, Protein(f"c{c}_{i}", None, doms)))
    key = operator.attrgetter("start")
    out_genes, contigs, ip, ap, at = native.sort_group(genes, key, index)
    assert all(a is b for a, b in zip(out_genes, genes)) and len(contigs) == 6
    for g in genes:
        starts = [d.start for d in g.protein.domains]
        assert starts == sorted(starts)
    ip2, ap2, at2 = native.pack_protein(contigs, index)
    assert (ip, ap, at) == (ip2, ap2, at2)
    py = packing.pack_contigs.__wrapped__(contigs, index) if hasattr(packing.pack_contigs, "__wrapped__") else None
    assert py is None or py.attr_id.tobytes() == at
    # the same arrays as the Python statements of features.py:31-35
    attr, aptr = [], [0]
    for g in genes:
        seen = []
        for d in g.protein.domains:
            if d.name not in seen:
                seen.append(d.name)
                if d.name in index:
                    attr.append(index[d.name])
        aptr.append(len(attr))
    assert np.frombuffer(at, dtype=np.int32).tolist() == attr and np.frombuffer(ap, dtype=np.int64).tolist() == aptr
    assert np.frombuffer(ip, dtype=np.int64).tolist() == np.cumsum([0] + [len(c) for c in contigs]).tolist()
    assert native.sort_group([], key, index) == ([], [], np.zeros(1, np.int64).tobytes(), np.zeros(1, np.int64).tobytes(), b"")

    # objects whose attributes are properties / slots: served through getattr, same result
    class SlotSource:
        __slots__ = ("_id",)

        def __init__(self, i):
            self._id = i

        @property
        def id(self):
            return self._id

    class PropGene:
        def __init__(self, g):
            self.__dict__["source"] = "shadowed"  # a data descriptor of the class wins over the instance dictionary
            self._g = g

        source = property(lambda self: SlotSource(self._g.source.id))
        start = property(lambda self: self._g.start)
        protein = property(lambda self: self._g.protein)

    wrapped = [PropGene(g) for g in genes]
    got = native.sort_group(wrapped, key, index)
    assert got is not None and (got[2], got[3], got[4]) == (ip, ap, at) and [len(c) for c in got[1]] == [len(c) for c in contigs]

    class Proxy:  # __getattribute__ of its own: the instance dictionary (which lies) must not be consulted
        def __init__(self, g):
            object.__setattr__(self, "_g", g)
            self.__dict__.update(start=-1, source=None, protein=None)

        def __getattribute__(self, name):
            if name in ("start", "source", "protein"):
                return getattr(object.__getattribute__(self, "_g"), name)
            return object.__getattribute__(self, name)

    got = native.sort_group([Proxy(g) for g in genes], key, index)
    assert got is not None and (got[2], got[3], got[4]) == (ip, ap, at)

    # annotate_all: list of floats == float64 buffer
    p = rng.random(len(genes))
    w1 = {nm: float(i) for i, nm in enumerate(names[:25])}
    a = native.annotate_all(genes, p.tolist(), w1, Gene, Protein, Domain)
    b = native.annotate_all(genes, p, w1, Gene, Protein, Domain)
    assert a == b and all(x is not y for x, y in zip(a, genes))
    assert [g._probability for g in b] == p.tolist() and all(type(g._probability) is float for g in b)
    assert all(d.probability == g._probability and d.cluster_weight == w1.get(d.name) for g in b for d in g.protein.domains)
    import pytest

    with pytest.raises(ValueError):
        native.annotate_all(genes, p[:-1], w1, Gene, Protein, Domain)
    with pytest.raises(ValueError):
        native.annotate_all(genes, p.astype(np.float32), w1, Gene, Protein, Domain)


# ---- the HIP runtime preload of gecco_amd._native (no blind preload of the torch wheel's libamdhip64) -------
def test_elf_reader_finds_the_hip_runtime_the_library_was_linked_against():
    from gecco_amd import _native

    needed = _native._elf_dynamic_strings(_native.LIB_PATH)[1]
    hip = [n for n in needed if n.startswith("libamdhip64")]
    assert len(hip) == 1 and hip[0].startswith("libamdhip64.so."), needed
    assert _native._elf_dynamic_strings(__file__) == {1: [], 14: []}  # (not an ELF file: nothing, no exception)


def test_hip_runtime_preload_is_gated_on_the_soname(monkeypatch):
    import sys
    import warnings

    from gecco_amd import _native

    monkeypatch.delitem(sys.modules, "torch", raising=False)
    monkeypatch.setenv("GECCO_AMD_HIP_RUNTIME", "system")
    assert _native._preload_hip_runtime(_native.LIB_PATH) is None
    monkeypatch.setenv("GECCO_AMD_HIP_RUNTIME", "auto")
    real = _native._elf_dynamic_strings

    def skewed(path, tags=(1, 14)):  # the wheel carries another ABI major than the one the library was linked against
        out = real(path, tags)
        if path != _native.LIB_PATH and out.get(14):
            out[14] = ["libamdhip64.so.6"]
        return out

    monkeypatch.setattr(_native, "_elf_dynamic_strings", skewed)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = _native._preload_hip_runtime(_native.LIB_PATH)
    import importlib.util

    if importlib.util.find_spec("torch") is not None:
        assert got is None
        assert any("libamdhip64.so.6" in str(c.message) and "keeping the system" in str(c.message) for c in caught)


def test_native_order_info_and_gather_equal_the_numpy_passes(monkeypatch):
    """`gecco_crf_packed_order_info` / `gecco_crf_gather_f64` (several host threads) against the numpy passes of predict_tables they
    replaced: gene rows in order or not, ties on (contig, start) with increasing / decreasing ends, many thread ranges."""
    from gecco_amd import _native as nat, predict, tables

    monkeypatch.setenv("GECCO_CRF_HOST_THREADS", "7")
    monkeypatch.setenv("GECCO_CRF_HOST_GRAIN", "1")
    model = nat.Model.from_tables(np.zeros((4, 2)), np.zeros((2, 2)))
    S = tables.StringColumn.from_sequence
    rng = np.random.default_rng(11)
    seen = set()
    for trial in range(60):
        nc = int(rng.integers(1, 6))
        sids, pids, starts, ends = [], [], [], []
        for c in range(nc):
            ng = int(rng.integers(1, 40))
            st = np.sort(rng.integers(0, 40 if trial % 3 else 10 ** 6, size=ng)) * 10  # (few distinct starts: ties)
            for i in range(ng):
                sids.append(f"ctg{c}")
                pids.append(f"c{c}_g{i}")
                starts.append(int(st[i]))
                ends.append(int(st[i] + rng.integers(1, 50)))
        n = len(pids)
        perm = np.arange(n) if trial % 2 == 0 else rng.permutation(n)  # gene table rows in scoring order or shuffled
        if trial % 4 == 1:  # stable sort by start keeps the row order among ties: make rows sorted but contigs interleaved
            perm = np.argsort(np.array(sids)[rng.permutation(n)], kind="stable")
        sids, pids = [sids[i] for i in perm], [pids[i] for i in perm]
        starts, ends = np.array(starts, dtype=np.int64)[perm], np.array(ends, dtype=np.int64)[perm]
        fi = np.sort(rng.integers(0, n, size=max(1, n)))
        pk = nat.PackedTables(model, S([sids[i] for i in fi]), S([pids[i] for i in fi]), starts[fi], S(["PF00001"] * len(fi)),
                              rng.integers(0, 300, size=len(fi)).astype(np.int64), S(sids), S(pids), starts)
        rows = pk.gene_row
        assert rows.min() >= 0 and pk.n_genes == n
        exp_in_order = bool(rows[0] == 0 and np.all(np.diff(rows) == 1))
        code = np.repeat(np.arange(pk.n_contigs), np.diff(pk.contig_ptr))
        exp_differs = predict._refiner_order_differs(code, starts[rows], ends[rows])
        assert pk.order_info(starts, ends) == (exp_in_order, exp_differs)
        seen.add((exp_in_order, exp_differs))
        p = rng.random(n)
        p[rng.random(n) < 0.1] = np.nan
        got = nat.gather_f64(p, pk.row_gene)
        np.testing.assert_array_equal(got, p[pk.row_gene])
    assert len(seen) == 4  # every combination occurred
    with pytest.raises(ValueError):
        nat.gather_f64(np.zeros(3), np.array([0, 3], dtype=np.int32))
    assert nat.gather_f64(np.zeros(3), np.zeros(0, dtype=np.int32)).shape == (0,)
