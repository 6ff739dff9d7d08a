"""Second-oracle checks: the C restatement vs brute-force path enumeration, plus the
wrapper semantics the fixture does not exercise (padding, pad=False, step>1)."""
import numpy as np
import pytest

from oracle import crf_oracle as orc
from tests.helpers import synth_contigs, synth_model


@pytest.mark.parametrize("L,T", [(2, 1), (2, 2), (2, 7), (2, 12), (3, 6), (4, 5)])
def test_marginals_vs_enumeration(L, T):
    rng = np.random.default_rng(100 * L + T)
    state = rng.normal(0, 2.0, size=(T, L))
    trans = rng.normal(0, 2.0, size=(L, L))
    marg, ln = orc.marginals_seq(state, trans)
    bm, bln = orc.brute_marginals(state, trans)
    np.testing.assert_allclose(marg, bm, rtol=0, atol=1e-13)
    assert abs(ln - bln) < 1e-11
    np.testing.assert_allclose(marg.sum(axis=1), 1.0, atol=1e-13)


@pytest.mark.parametrize("L,T", [(2, 1), (2, 9), (3, 6), (5, 4)])
def test_viterbi_vs_enumeration(L, T):
    rng = np.random.default_rng(7 * L + T)
    state = rng.normal(0, 2.0, size=(T, L))
    trans = rng.normal(0, 2.0, size=(L, L))
    lab, sc = orc.viterbi_seq(state, trans)
    bl, bs = orc.brute_viterbi(state, trans)
    assert lab.tolist() == bl.tolist()
    assert abs(sc - bs) < 1e-12


def test_viterbi_tie_breaks_to_first_label():
    # all-zero scores: every path ties; CRFsuite keeps the FIRST argmax (strict '<' update)
    lab, sc = orc.viterbi_seq(np.zeros((5, 3)), np.zeros((3, 3)))
    assert lab.tolist() == [0] * 5 and sc == 0.0


def _py_windowed(w, trans, cptr, gptr, attr, W, step, label, pad):
    """Literal python restatement of gecco/crf/__init__.py:209-258 on top of marginals_seq."""
    out = np.full(cptr[-1], np.nan)
    for c in range(len(cptr) - 1):
        g0, n = cptr[c], cptr[c + 1] - cptr[c]
        feats = [attr[gptr[g0 + g]:gptr[g0 + g + 1]] for g in range(n)]
        delta = 0
        if n < W:
            if not pad:
                continue
            delta = W - n
            feats = [[] for _ in range(delta // 2)] + feats + [[] for _ in range((delta + 1) // 2)]
        prob = np.zeros(max(n, W))
        for i in range(0, len(feats) + 1 - W, step):
            st = np.zeros((W, w.shape[1]))
            for t, f in enumerate(feats[i:i + W]):
                for a in f:
                    st[t] += w[a]
            m, _ = orc.marginals_seq(st, trans)
            np.maximum(prob[i:i + W], m[:, label], out=prob[i:i + W])
        out[g0:g0 + n] = prob[delta // 2:][:n]
    return out


@pytest.mark.parametrize("W,step,pad", [(20, 1, True), (20, 1, False), (5, 1, True), (5, 2, True), (7, 7, True), (1, 1, True)])
def test_windowed_wrapper_semantics(W, step, pad):
    rng = np.random.default_rng(W * 10 + step)
    w, trans = synth_model(50, rng)
    lengths = [1, 2, W - 1 if W > 1 else 1, W, W + 1, 3 * W + 2, 19, 20, 21, 57]
    cptr, gptr, attr = synth_contigs(rng, lengths, 50)
    got = orc.windowed_marginals(w, trans, cptr, gptr, attr, W, step, 1, pad)
    exp = _py_windowed(w, trans, cptr, gptr, attr, W, step, 1, pad)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(exp), rtol=0, atol=0)


def test_invalid_window_args():
    rng = np.random.default_rng(1)
    w, trans = synth_model(10, rng)
    cptr, gptr, attr = synth_contigs(rng, [5], 10)
    for W, step in [(0, 1), (5, 0), (5, 6)]:
        with pytest.raises(ValueError):
            orc.windowed_marginals(w, trans, cptr, gptr, attr, W, step)


def test_segment_semantics():
    nan = float("nan")
    #            0    1    2    3    4    5    6    7    8    9
    p = np.array([.9, .9, nan, .9, .1, nan, .85, .95, .81, .8])
    ann = np.array([0, 1, 1, 1, 1, 0, 1, 1, 1, 1], dtype=np.uint8)
    cptr = np.array([0, 10], dtype=np.int32)
    seg = orc.segment(p, ann, cptr, threshold=0.8, n_cds=3, edge_distance=0, trim=True)
    # run 0-3 (NaN inherits "in"), trimmed to 1-3 (gene 0 unannotated); gene 5 inherits "out";
    # run 6-8 (0.8 is NOT > 0.8)
    assert seg.tolist() == [[0, 1, 1, 4], [0, 2, 6, 9]]
    seg = orc.segment(p, ann, cptr, threshold=0.8, n_cds=3, edge_distance=0, trim=False)
    assert seg.tolist() == [[0, 1, 0, 4], [0, 2, 6, 9]]
    # edge distance 2: annotated ids = 1,2,3,4,6,7,8,9 -> edge = {1,2,8,9}
    seg = orc.segment(p, ann, cptr, threshold=0.8, n_cds=2, edge_distance=2, trim=True)
    assert seg.tolist() == [[0, 2, 6, 9]]  # cluster 1: {1,2,3}-edge={3} <2 ; cluster 2: {6,7,8}-edge={6,7}
    # grouper state leaks across contigs: second contig starts with NaN after an "in" gene
    p2 = np.array([.1, .9, nan, .9, .9])
    seg = orc.segment(p2, np.ones(5, dtype=np.uint8), np.array([0, 2, 5], dtype=np.int32), 0.8, 1, 0, True, carry_state=True)
    assert seg.tolist() == [[0, 1, 1, 2], [1, 1, 2, 5]]
    # ... unless every contig gets its own iter_clusters call, as in the CLI (_common.py:621-623)
    seg = orc.segment(p2, np.ones(5, dtype=np.uint8), np.array([0, 2, 5], dtype=np.int32), 0.8, 1, 0, True, carry_state=False)
    assert seg.tolist() == [[0, 1, 1, 2], [1, 1, 3, 5]]
    # a skipped (all-NaN) contig behind a contig that ends "in" is no cluster for the CLI
    p3 = np.array([.95] * 5 + [nan] * 5)
    cp3 = np.array([0, 5, 10], dtype=np.int32)
    assert len(orc.segment(p3, np.ones(10, dtype=np.uint8), cp3, 0.8, 3, 0, True, carry_state=False)) == 1
    assert len(orc.segment(p3, np.ones(10, dtype=np.uint8), cp3, 0.8, 3, 0, True, carry_state=True)) == 2
