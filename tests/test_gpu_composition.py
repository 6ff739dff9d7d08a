"""GPU parity of gecco_crf_domain_composition with the numpy oracle:
bit-exact, because the reference's arithmetic here is numpy.sum."""
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN, read_tsv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from gecco_amd import _native

    assert _native.device_count() >= 1
    return _native


def _random_case(rng, n_cols, n_genes, max_per_gene, col_pool):
    k = rng.integers(0, max_per_gene + 1, size=n_genes)
    dom_ptr = np.concatenate([[0], np.cumsum(k)]).astype(np.int32)
    rows = int(dom_ptr[-1])
    col = rng.choice(col_pool, size=rows).astype(np.int32)
    w = rng.random(rows) * 10.0 ** rng.integers(-6, 3, size=rows)  # wide range: summation order shows
    cuts = np.sort(rng.choice(np.arange(n_genes + 1), size=min(12, n_genes + 1), replace=False))
    seg = [(0, i + 1, int(a), int(b)) for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
    seg.append((0, 99, int(cuts[0]), int(cuts[0])))  # empty cluster
    return np.array(seg, dtype=np.int32), dom_ptr, col, w


@pytest.mark.parametrize("n_cols", [1, 5, 300, 1025, 2766, 8193, 20000, 140000])
def test_random_clusters_bit_exact(nat, n_cols):
    from oracle import composition as oc

    rng = np.random.default_rng(n_cols)
    pool = np.concatenate([[-1], rng.integers(0, n_cols, size=min(n_cols, 400))])
    seg, dom_ptr, col, w = _random_case(rng, n_cols, 300, 6, pool)
    for normalize in (True, False):
        got = nat.domain_composition(seg, dom_ptr, col, w, n_cols, normalize=normalize)
        exp = oc.compositions_packed(seg, dom_ptr, col, w, n_cols, normalize=normalize)
        assert np.array_equal(got, exp), (n_cols, normalize)


def test_many_copies_of_one_domain(nat):
    """>= 8 and > 128 terms per column exercise numpy's unrolled and recursive paths."""
    from oracle import composition as oc

    rng = np.random.default_rng(3)
    seg, dom_ptr, col, w = _random_case(rng, 7, 1500, 4, np.array([-1, 0, 1, 2, 6]))
    seg = np.array([(0, 1, 0, 1200), (0, 2, 1200, 1240), (0, 3, 1240, 1500)], dtype=np.int32)
    got = nat.domain_composition(seg, dom_ptr, col, w, 7)
    assert np.array_equal(got, oc.compositions_packed(seg, dom_ptr, col, w, 7))


def test_cluster_objects_match_reference_formula(nat):
    from gecco_amd import composition, model
    from oracle import composition as oc

    rng = np.random.default_rng(8)
    all_possible = [f"PF{i:05d}" for i in range(500)]
    clusters = []
    for c in range(6):
        genes = []
        for g in range(int(rng.integers(3, 30))):
            doms = [model.Domain(str(rng.choice(all_possible + ["TIGR0001"])), 1, 2, "Pfam", float(rng.random() * 1e-3),
                                 float(rng.random() * 1e-5)) for _ in range(int(rng.integers(0, 5)))]
            genes.append(model.Gene(model.Source("s"), 1, 10, model.Strand.Coding, model.Protein(f"c{c}_g{g}", None, doms)))
        clusters.append(model.Cluster(f"c{c}", genes))
    for kw in ({}, {"normalize": False}, {"minlog_weights": True}, {"pvalue": False}):
        got = composition.cluster_compositions(clusters, all_possible, **kw)
        for row, cl in zip(got, clusters):
            doms = [d for g in cl.genes for d in g.protein.domains]
            field = "pvalue" if kw.get("pvalue", True) else "i_evalue"
            import math
            w = [-math.log10(getattr(d, field)) if kw.get("minlog_weights") else 1 - getattr(d, field) for d in doms]
            exp = oc.domain_composition([d.name for d in doms], w, all_possible, normalize=kw.get("normalize", True))
            assert np.array_equal(row, exp), kw
    one = composition.domain_composition(clusters[0])
    doms = [d for g in clusters[0].genes for d in g.protein.domains]
    assert np.array_equal(one, oc.domain_composition([d.name for d in doms], [1 - d.pvalue for d in doms]))


def test_golden_cluster_through_columnar_predict(nat, oracle_model):
    from gecco_amd import predict, tables
    from gecco_amd.crf import ClusterCRF
    from oracle import composition as oc

    crf = ClusterCRF.trained(GOLDEN)
    genes_t = tables.GeneTable.load(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    feats_t = tables.FeatureTable.load(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    all_possible = sorted(oracle_model["attr_index"])
    _, _, clusters, comps = predict.predict_tables(genes_t, feats_t, crf, composition_domains=all_possible)
    assert len(clusters) == 1 and comps.shape == (1, len(all_possible))
    feats = read_tsv(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    by_gene = {}
    for r in feats:
        by_gene.setdefault(r["protein_id"], []).append(r)
    members = set(clusters.proteins[0].split(";"))
    rows = [r for pid in genes_t.protein_id if pid in members  # genes in contig order, domains by start
            for r in sorted(by_gene.get(pid, []), key=lambda r: int(r["domain_start"]))]
    exp = oc.domain_composition([r["domain"] for r in rows], [1 - float(r["pvalue"]) for r in rows], all_possible)
    assert np.array_equal(comps[0], exp)


def test_no_clusters_and_no_domains(nat):
    out = nat.domain_composition(np.zeros((0, 4), dtype=np.int32), [0, 1, 2], [3, 4], [0.5, 0.25], 10)
    assert out.shape == (0, 10)
    out = nat.domain_composition(np.array([(0, 1, 0, 3)], dtype=np.int32), [0, 0, 0, 0], [], [], 10)
    assert out.shape == (1, 10) and not out.any()
