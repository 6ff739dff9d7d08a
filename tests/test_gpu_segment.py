"""GPU parity of row R on packed arrays: gecco_crf_segment vs the oracle's restatement of
GeneGrouper / ClusterRefiner (gecco/refine.py:51-64,118-200)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_case(rng, n_contigs, max_len):
    p_all, ann_all, cptr = [], [], [0]
    for c in range(n_contigs):
        n = int(rng.integers(0, max_len))
        mode = rng.random()
        p = np.clip(rng.normal(0.9 if mode < 0.4 else 0.3, 0.3, size=n), 0, 1)
        p[rng.random(n) < 0.08] = np.nan
        if rng.random() < 0.1:
            p[:] = np.nan  # a contig without any prediction inherits the grouper state
        p_all.append(p)
        ann_all.append(rng.random(n) < 0.65)
        cptr.append(cptr[-1] + n)
    return np.concatenate(p_all), np.concatenate(ann_all).astype(np.uint8), np.array(cptr, dtype=np.int32)


@pytest.mark.parametrize("n_cds,edge,trim", [(3, 0, True), (1, 0, False), (2, 2, True), (5, 1, True), (1, 10, True)])
def test_segment_matches_oracle(n_cds, edge, trim):
    from gecco_amd import _native as nat
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(100 * n_cds + edge)
    for n_contigs, max_len in [(1, 50), (40, 80), (3000, 60), (5, 5000), (3, 40000)]:
        p, ann, cptr = _random_case(rng, n_contigs, max_len)
        for carry in (False, True):  # one grouper per contig (the CLI) / one for the whole call
            exp = orc.segment(p, ann, cptr, 0.8, n_cds, edge, trim, carry_state=carry)
            got = nat.segment(p, ann, cptr, 0.8, n_cds, edge, trim, carry_state=carry)
            assert got.tolist() == exp.tolist()


def test_segment_golden(oracle_model):
    from gecco_amd import _native as nat
    from oracle import crf_oracle as orc
    from tests.helpers import golden_csr

    ids, cptr, gptr, attr, expected, ann = golden_csr(oracle_model["attr_index"])
    seg = nat.segment(expected, ann, cptr, 0.8, 3, 0, True)
    assert seg.tolist() == [[0, 1, 0, 23]]
    assert nat.segment(np.zeros(0), np.zeros(0, dtype=np.uint8), [0]).shape == (0, 4)


def test_segment_long_runs_and_empty_contigs():
    """Runs that span many scan blocks, contigs without genes, runs without any annotated gene."""
    from gecco_amd import _native as nat
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(5)
    n = 30000
    p = np.full(n, 0.95)
    p[rng.integers(0, n, size=6)] = 0.1          # a handful of cuts: runs of thousands of genes
    p[20000:20100] = np.nan
    ann = (rng.random(n) < 0.5).astype(np.uint8)
    ann[:3000] = 0                                # the first run trims to nothing
    cptr = np.array([0, 0, 12000, 12000, 12001, n, n], dtype=np.int32)
    for n_cds, edge, trim in [(3, 0, True), (1, 5, True), (0, 0, True), (2, 0, False), (3, 100000, True)]:
        for carry in (False, True):
            exp = orc.segment(p, ann, cptr, 0.8, n_cds, edge, trim, carry_state=carry)
            got = nat.segment(p, ann, cptr, 0.8, n_cds, edge, trim, carry_state=carry)
            assert got.tolist() == exp.tolist(), (n_cds, edge, trim, carry)
