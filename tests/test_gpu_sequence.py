"""GPU parity of the whole-contig extensions (rows F and V): full-sequence marginals and
Viterbi decoding vs the CPU oracle ([EXT] CRFsuite semantics)."""
import numpy as np
import pytest

from tests.helpers import golden_csr, synth_contigs, synth_model

pytestmark = pytest.mark.gpu

LENGTHS = [1, 2, 3, 63, 64, 65, 127, 128, 129, 200, 1000, 4097, 20000]


@pytest.fixture(scope="module")
def nat():
    from gecco_amd import _native

    assert _native.device_count() >= 1
    return _native


@pytest.fixture(scope="module")
def real(nat, oracle_model):
    import os

    from oracle import lcrf
    from tests.helpers import GOLDEN

    return nat.Model.from_lcrf(lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))["blob"])


def test_golden_contig(real, oracle_model):
    from oracle import crf_oracle as orc

    ids, cptr, gptr, attr, expected, ann = golden_csr(oracle_model["attr_index"])
    marg, ln = real.marginals_full(cptr, gptr, attr)
    emarg, eln = orc.full_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.abs(marg - emarg).max() <= 1e-12 and abs(ln[0] - eln[0]) <= 1e-9 * abs(eln[0])
    y, sc = real.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert y.tolist() == ey.tolist() == [1] * 23 and abs(sc[0] - esc[0]) <= 1e-9


def test_full_marginals_random(real, oracle_model):
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(21)
    cptr, gptr, attr = synth_contigs(rng, LENGTHS + list(rng.integers(1, 300, size=50)), oracle_model["state"].shape[0])
    marg, ln = real.marginals_full(cptr, gptr, attr)
    emarg, eln = orc.full_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.abs(marg - emarg).max() <= 1e-12
    np.testing.assert_allclose(marg.sum(axis=1), 1.0, atol=1e-14)
    assert np.abs(ln - eln).max() <= 1e-9 * np.abs(eln).max()


def test_viterbi_random(real, oracle_model):
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(22)
    cptr, gptr, attr = synth_contigs(rng, LENGTHS + list(rng.integers(1, 300, size=50)), oracle_model["state"].shape[0])
    y, sc = real.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.array_equal(y, ey.astype(np.int8))
    assert np.abs(sc - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_viterbi_exact_ties_integer_weights(nat, seed):
    """Integer-valued weights: every addition is exact, so ties are real ties and CRFsuite's
    first-argmax rule must be reproduced exactly, also across chunk boundaries."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(seed)
    A = 6
    w = rng.integers(-1, 2, size=(A, 2)).astype(float)
    trans = rng.integers(-1, 2, size=(2, 2)).astype(float) if seed else np.zeros((2, 2))
    cptr, gptr, attr = synth_contigs(rng, [1, 5, 64, 65, 130, 700, 3000], A)
    model = nat.Model.from_tables(w, trans)
    y, sc = model.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y, ey.astype(np.int8))
    assert np.array_equal(sc, esc)


def test_full_marginals_very_long_contigs(real, oracle_model):
    """Contigs spanning > 64 workgroups: the look-back over forward totals and the look-ahead over
    backward totals both take more than one 64-wide step; also contig ends on workgroup borders."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(78)
    cptr, gptr, attr = synth_contigs(rng, [300000, 3, 140000, 2049, 2048, 2048, 1, 4095], oracle_model["state"].shape[0])
    marg, ln = real.marginals_full(cptr, gptr, attr)
    emarg, eln = orc.full_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.abs(marg - emarg).max() <= 1e-12
    assert np.abs(ln - eln).max() <= 1e-9 * np.abs(eln).max()


@pytest.mark.parametrize("integer_weights", [False, True])
def test_viterbi_very_long_contigs_walk_back_many_workgroups(nat, real, oracle_model, integer_weights):
    """Contigs spanning > 64 workgroups of 2048 genes: the look-back over workgroup totals (prefix
    scores) and the look-ahead over workgroup label maps both take more than one 64-wide step."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(77)
    if integer_weights:  # label-neutral genes dominate: the two paths stay apart over long stretches
        A = 8
        w = np.zeros((A, 2))
        w[0] = (1.0, 0.0)
        w[1] = (0.0, 1.0)
        trans = np.array([[1.0, -3.0], [-3.0, 1.0]])
        model = nat.Model.from_tables(w, trans)
    else:
        A = oracle_model["state"].shape[0]
        w, trans, model = oracle_model["state"], oracle_model["trans"], real
    cptr, gptr, attr = synth_contigs(rng, [300000, 3, 140000, 2049], A)
    y, sc = model.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y, ey.astype(np.int8))
    assert np.abs(sc - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_viterbi_difference_form(nat, real, oracle_model, seed, monkeypatch):
    """Labels without path scores take the score-difference form of the recursion (8 B/gene): same
    labels as the oracle and as the matrix form, on the shipped model, on integer weights (exact
    ties) and on very long contigs."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(200 + seed)
    if seed == 0:
        w, trans, model = oracle_model["state"], oracle_model["trans"], real
        lengths = LENGTHS + list(rng.integers(1, 300, size=50))
    elif seed == 1:
        w = rng.integers(-1, 2, size=(6, 2)).astype(float)
        trans = np.array([[1.0, -1.0], [-1.0, 1.0]])
        model = nat.Model.from_tables(w, trans)
        lengths = [1, 5, 64, 65, 130, 700, 3000, 2048, 2049]
    elif seed == 2:
        w = np.zeros((8, 2))
        w[0], w[1] = (1.0, 0.0), (0.0, 1.0)
        trans = np.array([[1.0, -3.0], [-3.0, 1.0]])
        model = nat.Model.from_tables(w, trans)
        lengths = [300000, 3, 140000, 2049]
    else:
        w, _ = synth_model(500, rng)
        trans = np.array([[0.3, -0.2], [0.1, 0.25]])  # lo = -0.45 <= hi = 0.2, barely sticky
        model = nat.Model.from_tables(w, trans)
        lengths = list(rng.integers(1, 3000, size=40))
    cptr, gptr, attr = synth_contigs(rng, lengths, w.shape[0])
    y, sc = model.viterbi(cptr, gptr, attr, want_score=False)
    assert sc is None
    ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y, ey.astype(np.int8))
    monkeypatch.setenv("GECCO_CRF_VITERBI", "matrix")
    ym, _ = model.viterbi(cptr, gptr, attr, want_score=False)
    assert np.array_equal(y, ym)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_viterbi_short_contigs_single_kernel(nat, real, oracle_model, seed):
    """No contig longer than one 2048-gene scan block: fold and replay run as one kernel that looks
    back by recomputing the stretch between the contig start and the block.  Contig boundaries are
    placed on, just before and just after block boundaries."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(400 + seed)
    if seed == 0:
        w, trans, model = oracle_model["state"], oracle_model["trans"], real
        lengths = [2048, 2048, 1, 2047, 2048, 5, 2043, 2048, 2000, 48, 1, 1, 2046] + list(rng.integers(1, 2049, size=60))
    elif seed == 1:
        w = rng.integers(-1, 2, size=(6, 2)).astype(float)
        trans = np.array([[1.0, -1.0], [-1.0, 1.0]])
        model = nat.Model.from_tables(w, trans)
        lengths = [2048, 3, 2045, 2048, 1024, 1024, 1, 2047] + list(rng.integers(1, 2049, size=30))
    else:
        w = np.zeros((8, 2))
        w[0], w[1] = (1.0, 0.0), (0.0, 1.0)
        trans = np.array([[1.0, -3.0], [-3.0, 1.0]])
        model = nat.Model.from_tables(w, trans)
        lengths = list(rng.integers(1500, 2049, size=40))
    cptr, gptr, attr = synth_contigs(rng, lengths, w.shape[0])
    y, _ = model.viterbi(cptr, gptr, attr, want_score=False)
    ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y, ey.astype(np.int8))


def test_viterbi_antisticky_model_falls_back_to_matrix_form(nat):
    """t01 - t11 > t00 - t10: the difference recursion is not a clamp; the general form must be used."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(31)
    w, _ = synth_model(300, rng)
    trans = np.array([[-1.0, 2.0], [1.5, -0.5]])
    cptr, gptr, attr = synth_contigs(rng, [1, 2, 70, 500, 4100], 300)
    y, _ = nat.Model.from_tables(w, trans).viterbi(cptr, gptr, attr, want_score=False)
    ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y, ey.astype(np.int8))


def test_synthetic_model_c2_shape(nat):
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(5)
    A = 35000
    w, trans = synth_model(A, rng)
    lengths = np.clip(np.round(rng.lognormal(np.log(200), 0.5, size=300)), 5, 2000).astype(int)
    cptr, gptr, attr = synth_contigs(rng, lengths, A)
    model = nat.Model.from_tables(w, trans)
    marg, ln = model.marginals_full(cptr, gptr, attr)
    emarg, eln = orc.full_marginals(w, trans, cptr, gptr, attr)
    assert np.abs(marg - emarg).max() <= 1e-12
    y, sc = model.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y, ey.astype(np.int8))


def test_model_view_single_sequence_api(oracle_model):
    """[EXT] CRF.predict_marginals_single / predict_single surface of ClusterCRF.model."""
    from gecco_amd.crf import ClusterCRF
    from oracle import crf_oracle as orc
    from tests.helpers import GOLDEN

    crf = ClusterCRF.trained(GOLDEN)
    xseq = [{"PF00109": True, "PF02801": True}, {}, {"PF08659": True, "nope": True}, {"PF00106": True}]
    marg = crf.model.predict_marginals_single(xseq)
    ai = oracle_model["attr_index"]
    gptr = [0, 2, 2, 3, 4]
    attr = [ai["PF00109"], ai["PF02801"], ai["PF08659"], ai["PF00106"]]
    emarg, _ = orc.full_marginals(oracle_model["state"], oracle_model["trans"], [0, 4], gptr, attr)
    assert all(abs(m["1"] - e[1]) <= 1e-12 and abs(m["0"] - e[0]) <= 1e-12 for m, e in zip(marg, emarg))
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], [0, 4], gptr, attr)
    assert crf.model.predict_single(xseq) == [str(v) for v in ey.tolist()]
    assert crf.model.predict_marginals_single([]) == [] and crf.model.predict_single([]) == []


def test_empty_inputs(real):
    marg, ln = real.marginals_full([0], [0], [])
    assert marg.shape == (0, 2) and ln.shape == (0,)
    y, sc = real.viterbi([0, 0, 3], [0, 1, 1, 2], [5, 7])
    assert y.shape == (3,) and sc[0] == 0.0


def test_viterbi_short_contigs_reproduce_the_sequential_difference_recursion(nat):
    """Decisions must not depend on how the scan is cut.  Values entering a lane come from COMPOSED clamp maps,
    whose additions are associated differently from the sequential recursion; `vd_short` therefore rebuilds
    every entering value sequentially from the last position where the clamp had forgotten the past.  The
    contigs below never saturate (transitions +-1000) and end on a score difference of EXACTLY zero in the
    sequential recursion -- one ulp of re-association noise flips the end label and with it every label of the
    contig.  Checker: oracle.viterbi_delta, the strictly sequential recursion (and CRFsuite's own form, which
    sees the same exact tie)."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(77)
    lengths = [int(x) for x in rng.integers(17, 400, size=600)]
    n = sum(lengths)
    trans = np.array([[0.0, -1000.0], [-1000.0, 0.0]])
    w = np.zeros((n, 2))
    gptr = np.arange(n + 1, dtype=np.int32)          # gene g carries attribute g alone: d_g = w[g][1]
    attr = np.arange(n, dtype=np.int32)
    cptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    sensitive = 0
    for c, T in enumerate(lengths):
        d = rng.uniform(-1.0, 1.0, size=T)
        D = d[0]
        for t in range(1, T - 1):
            D = D + (0.0 + d[t])                      # the recursion without saturation, sequentially
        d[T - 1] = -D                                  # ... ends on exactly 0: label 0 (strict >)
        w[cptr[c]:cptr[c + 1], 1] = d
        # would a scan that adds the genes of lanes 1.. first, then the first lane, see it differently?
        grouped = d[0] + (np.sum(d[8:T - 1]) if T > 9 else 0.0) + np.sum(d[1:8])
        sensitive += int(grouped + d[T - 1] != 0.0)
    assert sensitive > 50  # the case is real: re-association changes the last bits for many contigs
    model = nat.Model.from_tables(w, trans)
    exp = orc.viterbi_delta(w, trans, cptr, gptr, attr)
    assert int(exp.sum()) == 0
    y, _ = model.viterbi(cptr, gptr, attr, want_score=False)
    np.testing.assert_array_equal(y.astype(np.int32), exp)
    ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)   # CRFsuite's form: delta[1] accumulates the same sums
    np.testing.assert_array_equal(ey, exp)
    # and through the decode path (state differences written by the windowed kernel)
    ses = nat.Session(model, [0])
    _, y2 = ses.decode(cptr, gptr, attr, 20)
    np.testing.assert_array_equal(y2.astype(np.int32), exp)


@pytest.mark.parametrize("seed", [0, 1])
def test_viterbi_short_contigs_equal_sequential_recursion_on_random_models(nat, seed):
    from oracle import crf_oracle as orc
    from tests.helpers import synth_contigs, synth_model

    rng = np.random.default_rng(900 + seed)
    A = 500
    w, trans = synth_model(A, rng)
    cptr, gptr, attr = synth_contigs(rng, list(rng.integers(1, 2048, size=80)) + [2048, 2047, 1, 9, 8], A)
    y, _ = nat.Model.from_tables(w, trans).viterbi(cptr, gptr, attr, want_score=False)
    np.testing.assert_array_equal(y.astype(np.int32), orc.viterbi_delta(w, trans, cptr, gptr, attr))


def test_random_shapes_against_the_oracle(monkeypatch, capsys):
    """tools/stress_sequence.py: 80 random batches whose workgroups of whole contigs end at, just before and just after
    2048 genes, partial last lanes, one-gene contigs, a long contig among short ones -- labels equal, marginals and log Z
    within 1e-12 of the oracle."""
    import os
    import runpy
    import sys

    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_sequence.py")
    monkeypatch.setattr(sys, "argv", [tool, "80", "7"])
    runpy.run_path(tool, run_name="__main__")
    assert "ok 80 batches" in capsys.readouterr().out


def _plant_end_tie(rng, T, w, trans, base_attr, ulps):
    """A contig of T genes (gene g carries attribute base_attr + g alone) whose CRFsuite delta recursion ends on
    delta[1] - delta[0] = `ulps` units in the last place of delta[0]: 0 = an exact tie (first arg max: label 0),
    +1 = label 1 by one ulp.  Real-valued weights everywhere else; returns nothing (fills w in place)."""
    t00, t01, t10, t11 = trans[0, 0], trans[0, 1], trans[1, 0], trans[1, 1]
    w[base_attr:base_attr + T, 0] = rng.normal(0.0, 1.0, size=T)
    w[base_attr:base_attr + T, 1] = rng.normal(0.0, 1.0, size=T)
    d0, d1 = w[base_attr, 0], w[base_attr, 1]
    for t in range(1, T - 1):  # [EXT] crf1dc_viterbi, the oracle's oracle_viterbi_seq
        a0, b0, a1, b1 = d0 + t00, d1 + t10, d0 + t01, d1 + t11
        m0 = b0 if a0 < b0 else a0
        m1 = b1 if a1 < b1 else a1
        d0, d1 = m0 + w[base_attr + t, 0], m1 + w[base_attr + t, 1]
    a0, b0, a1, b1 = d0 + t00, d1 + t10, d0 + t01, d1 + t11
    m0 = b0 if a0 < b0 else a0
    m1 = b1 if a1 < b1 else a1
    # last gene: s0 = 0, s1 chosen so that m1 + s1 lands exactly `ulps` ulps above m0 + 0
    target = m0
    for _ in range(abs(ulps)):
        target = np.nextafter(target, np.inf if ulps > 0 else -np.inf)
    s1 = target - m1
    for _ in range(64):  # the subtraction rounds: walk s1 until the sum is exact
        got = m1 + s1
        if got == target:
            break
        s1 = np.nextafter(s1, np.inf if got < target else -np.inf)
    assert m1 + s1 == target
    w[base_attr + T - 1] = (0.0, s1)


def test_viterbi_labels_are_crfsuites_on_planted_ties(nat):
    """Labels are CRFsuite's, not merely those of an equivalent recursion: contigs whose decisions keep their distance
    from the thresholds are decided identically by every form (the rounding noise of any of them is far below the
    margin), and a contig with a decision inside the margin is decoded again with CRFsuite's own delta recursion
    (exact_delta_contig).  Here: real-valued weights, contigs of 200 (one launch) and of 50 000 genes (look-back path)
    that end on an exact tie of the ACCUMULATED scores or one ulp either side of it -- the difference form sees another
    rounding of the same quantity and, on its own, gets a share of them wrong.  Checker: oracle.viterbi, the delta form."""
    from oracle import crf_oracle as orc

    trans = np.array([[2.669891070463728, -2.599571900486168], [-2.6019205422130995, 2.5683226020688488]])
    import torch

    for lengths in ([200] * 300, [50000, 177, 50000, 50000, 23, 50000], [300000, 61, 300000]):
        rng = np.random.default_rng(len(lengths))
        n = sum(lengths)
        w = np.zeros((n, 2))
        cptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
        gptr = np.arange(n + 1, dtype=np.int32)
        attr = np.arange(n, dtype=np.int32)
        for c, T in enumerate(lengths):
            _plant_end_tie(rng, T, w, trans, int(cptr[c]), ulps=(0, 1, -1)[c % 3])
        model = nat.Model.from_tables(w, trans)
        ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)
        # the planted ends decide the last label: tie -> 0, +1 ulp -> 1, -1 ulp -> 0
        np.testing.assert_array_equal(ey[cptr[1:] - 1], [(0, 1, 0)[c % 3] for c in range(len(lengths))])
        y, _ = model.viterbi(cptr, gptr, attr, want_score=False)
        np.testing.assert_array_equal(y.astype(np.int32), ey)
        ses = nat.Session(model, [0])
        _, y2 = ses.decode(cptr, gptr, attr, 20)
        np.testing.assert_array_equal(y2.astype(np.int32), ey)
        # the difference form alone does not get all of them (the case is real) -- unless every planted end happens to
        # round the same way in both forms
        yd = orc.viterbi_delta(w, trans, cptr, gptr, attr)
        assert (yd != ey).any() or len(lengths) < 10
        # what the decoder says it did: every planted end lies inside the margin of its distance to the last saturation,
        # so every contig went through CRFsuite's recursion -- and nothing else did (real-valued weights elsewhere)
        plan = nat.Plan(model, cptr, 20, 1, True, device=0)
        d_gp, d_at = torch.from_numpy(gptr).cuda(), torch.from_numpy(attr).cuda()
        d_y = torch.zeros(n, dtype=torch.int8, device="cuda:0")
        plan.viterbi_stats(reset=True)
        plan.run_viterbi(d_gp.data_ptr(), d_at.data_ptr(), d_y.data_ptr())
        st = plan.viterbi_stats()
        np.testing.assert_array_equal(d_y.cpu().numpy().astype(np.int32), ey)
        assert st["contigs_redecoded"] == len(lengths) and st["genes_redecoded"] == n, st
        assert len(lengths) <= st["inside_margin"] <= st["candidates"], st


def test_viterbi_margin_is_rigorous_and_rarely_met(nat):
    """The margin inside which a decision of the difference form is not provably CRFsuite's is (4 r + 4) ulp(M), r = genes
    since both labels last shared a predecessor for certain, M = bound on the contig's accumulated scores
    (crf_vd_short.hpp).  On real-valued models it is practically never met: no contig of a C2-shaped batch, nor of five
    30 000-gene contigs, is decoded again -- and the labels are the oracle's delta recursion's (= [EXT] crf1dc_viterbi).
    A model with small integer weights ties exactly all the time: there the second pass does run."""
    import torch
    from oracle import crf_oracle as orc
    from tests.helpers import synth_contigs, synth_model

    rng = np.random.default_rng(4242)
    A = 3000
    w, trans = synth_model(A, rng)
    for lengths in (list(np.clip(np.round(rng.lognormal(np.log(200), 0.5, size=400)), 5, 2000).astype(int)), [30000] * 5):
        cptr, gptr, attr = synth_contigs(rng, lengths, A)
        n = int(cptr[-1])
        ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)
        for ww, expect_redecode in ((w, False), (np.round(w), True)):
            model = nat.Model.from_tables(ww, np.round(trans) if expect_redecode else trans)
            plan = nat.Plan(model, cptr, 20, 1, True, device=0)
            d_gp, d_at = torch.from_numpy(gptr).cuda(), torch.from_numpy(attr).cuda()
            d_y = torch.zeros(n, dtype=torch.int8, device="cuda:0")
            plan.viterbi_stats(reset=True)
            plan.run_viterbi(d_gp.data_ptr(), d_at.data_ptr(), d_y.data_ptr())
            st = plan.viterbi_stats()
            if expect_redecode:
                ey2, _ = orc.viterbi(np.round(w), np.round(trans), cptr, gptr, attr)
                np.testing.assert_array_equal(d_y.cpu().numpy().astype(np.int32), ey2)
                assert st["contigs_redecoded"] > 0, st
            else:
                np.testing.assert_array_equal(d_y.cpu().numpy().astype(np.int32), ey)
                assert st["contigs_redecoded"] == 0 and st["inside_margin"] == 0, st
