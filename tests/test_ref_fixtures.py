"""Host logic and the oracle against vectors produced by THE REFERENCE'S OWN PYTHON (tests/golden/ref_*.json.gz, written in the
build container by tools/gen_reference_fixtures.py, which imports /root/reference/gecco and runs
`ClusterCRF.predict_probabilities`, `ClusterRefiner.iter_clusters`, `Cluster.domain_composition`; only its tagger is the C
oracle).  CPU only.  These pin rows D / X / W (pad, skip, windows, step, max), N (order, cluster_weight), R (refiner) and f4
(composition) of SURVEY.md 8a on reference-produced data instead of on this repository's reading of the reference."""
import itertools
import operator

import numpy as np
import pytest

from tests.helpers import genes_from_crf_case, genes_from_refiner_case, load_ref, pack_refiner_case


@pytest.fixture(scope="module")
def crf_cases():
    return load_ref("ref_predict_probabilities")


@pytest.fixture(scope="module")
def refiner_cases():
    return load_ref("ref_refiner")


def _sorted_contigs(genes):
    genes = sorted(genes, key=operator.attrgetter("source.id", "start"))
    for g in genes:
        g.protein.domains.sort(key=operator.attrgetter("start"))
    return genes, [list(g) for _, g in itertools.groupby(genes, key=operator.attrgetter("source.id"))]


def test_fixture_shape(crf_cases, refiner_cases):
    assert len(crf_cases) >= 200 and len(refiner_cases) >= 200
    ok = [c for c in crf_cases if "expect" in c]
    assert len(ok) >= 200
    params = {(c["params"]["window_size"], c["params"]["window_step"], c["params"]["pad"], c["params"]["feature_type"]) for c in ok}
    assert {w for w, _, _, _ in params} == {5, 20, 32} and {s for _, s, _, _ in params} >= {1, 3, 20}
    assert any(f == "domain" for _, _, _, f in params)
    assert any(c["warnings"] for c in ok) and sum(len(c.get("clusters", [])) for c in refiner_cases) > 300


def test_oracle_pad_window_step_max_equals_the_reference_loop(crf_cases, oracle_model):
    """Protein mode: the host packer (row X) + the oracle's pad / sliding-window / max wrapper (rows D, W) give, BIT FOR BIT, what
    the reference's own loop (crf/__init__.py:209-258, _meta.py:124-132) made of the same tagger: pad True / False, step 1 / 3 /
    20, windows of 5 / 20 / 32, contigs of W - 1, W, W + 1 genes, unknown domains, repeated domains, unsorted input."""
    from gecco_amd import packing
    from oracle import crf_oracle as orc

    n = 0
    for case in crf_cases:
        prm = case["params"]
        if prm["feature_type"] != "protein" or "expect" not in case:
            continue
        genes, contigs = _sorted_contigs(genes_from_crf_case(case))
        assert [g.protein.id for g in genes] == case["expect"]["order"]
        batch = packing.pack_contigs(contigs, oracle_model["attr_index"], "protein")
        got = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], batch.item_ptr.astype(np.int32), batch.attr_ptr.astype(np.int32),
                                     batch.attr_id, prm["window_size"], prm["window_step"], 1, prm["pad"])
        exp = np.array([np.nan if p is None else p for p in case["expect"]["p"]], dtype=np.float64)
        assert got.shape == exp.shape
        assert np.array_equal(np.isnan(got), np.isnan(exp))  # pad=False: skipped contigs keep no probability
        assert got[~np.isnan(got)].tobytes() == exp[~np.isnan(exp)].tobytes()
        n += 1
    assert n >= 180


def test_refiner_class_equals_the_reference(refiner_cases):
    """`gecco_amd.refine.ClusterRefiner.iter_clusters` (one call over all contigs: the grouper's state carries over) against
    the reference's: ids, members, average_probability (statistics.mean: exactly rounded), maximum, start, end -- both
    criteria, edge distance 0-3, trim on / off, genes without probability, equal starts, unsorted input."""
    from gecco_amd.refine import ClusterRefiner

    n = 0
    for case in refiner_cases:
        genes = genes_from_refiner_case(case)
        refiner = ClusterRefiner(**case["params"])
        if "error" in case:  # (antismash criterion over a gene without probability: TypeError out of numpy.mean, here too)
            with pytest.raises(TypeError):
                list(refiner.iter_clusters(genes))
            continue
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = [[c.id, [g.protein.id for g in c.genes], c.average_probability, c.maximum_probability, c.start, c.end]
                   for c in refiner.iter_clusters(genes)]
        assert got == case["clusters"]
        n += len(got)
    assert n > 300


def test_oracle_segmenter_equals_the_reference(refiner_cases):
    """The oracle's packed-array restatement of the grouper + refiner (oracle/crf_oracle.c oracle_segment*, the checker of
    csrc/crf_segment.hip) against the reference's clusters: rows (contig, number, first, last + 1) name the same genes."""
    from gecco_amd.refine import BIO_PFAMS
    from oracle import crf_oracle as orc

    markers = sorted(BIO_PFAMS)
    n = 0
    for case in refiner_cases:
        if "error" in case:
            continue
        prm = case["params"]
        ids, cids, p, ann, cptr, mptr, mid = pack_refiner_case(case, markers)
        if prm["criterion"] == "gecco":
            seg = orc.segment(p, ann, cptr, prm["threshold"], prm["n_cds"], prm["edge_distance"], prm["trim"], carry_state=True)
        else:
            seg = orc.segment_antismash(p, ann, cptr, mptr, mid, prm["threshold"], prm["n_cds"], prm["n_biopfams"], prm["average_threshold"],
                                        prm["trim"], carry_state=True)
        got = [[f"{cids[c]}_cluster_{k}", ids[a:b]] for c, k, a, b in seg.tolist()]
        assert got == [[c[0], c[1]] for c in case["clusters"]]
        n += len(got)
    assert n > 300


def test_oracle_composition_equals_the_reference():
    """oracle/composition.py (the checker of csrc/crf_composition.hip) against `Cluster.domain_composition` of the reference,
    bit for bit: all_possible given / None, normalised or not, p-values or e-values."""
    from oracle import composition as oc

    for case in load_ref("ref_composition"):
        names = [d[0] for g in case["genes"] for d in g]
        for key, exp in case["composition"].items():
            normalize, pvalue = key.split(",")[0].endswith("1"), key.split(",")[1].endswith("1")
            weights = [1 - (d[2] if pvalue else d[1]) for g in case["genes"] for d in g]
            got = oc.domain_composition(names, weights, case["all_possible"], normalize=normalize)
            assert np.asarray(got, dtype=np.float64).tobytes() == np.asarray(exp, dtype=np.float64).tobytes()


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/gecco"), reason="build container only: needs the reference's sources")
def test_committed_fixtures_are_what_the_generator_writes_today(tmp_path):
    """Where the reference is present (the build container), `tools/gen_reference_fixtures.py` is run again and must write the
    committed files byte for byte: the vectors under tests/golden/ ARE the reference's outputs, not an edited copy."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cp = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_reference_fixtures.py"), "--out", str(tmp_path)], capture_output=True,
                        text=True, timeout=600, cwd=root)
    assert cp.returncode == 0, cp.stderr[-2000:]
    for name in ("ref_predict_probabilities", "ref_refiner", "ref_composition"):
        with open(tmp_path / f"{name}.json.gz", "rb") as a, open(os.path.join(root, "tests", "golden", f"{name}.json.gz"), "rb") as b:
            assert a.read() == b.read(), name
