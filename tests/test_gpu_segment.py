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
    for n_contigs, max_len in [(1, 50), (40, 80), (3000, 60), (5, 5000)]:
        p, ann, cptr = _random_case(rng, n_contigs, max_len)
        exp = orc.segment(p, ann, cptr, 0.8, n_cds, edge, trim)
        got = nat.segment(p, ann, cptr, 0.8, n_cds, edge, trim)
        assert got.tolist() == exp.tolist()


def test_segment_golden(oracle_model):
    from gecco_amd import _native as nat
    from oracle import crf_oracle as orc
    from tests.helpers import golden_csr

    ids, cptr, gptr, attr, expected, ann = golden_csr(oracle_model["attr_index"])
    seg = nat.segment(expected, ann, cptr, 0.8, 3, 0, True)
    assert seg.tolist() == [[0, 1, 0, 23]]
    assert nat.segment(np.zeros(0), np.zeros(0, dtype=np.uint8), [0]).shape == (0, 4)
