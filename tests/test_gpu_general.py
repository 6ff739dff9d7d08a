"""GPU parity of the any-number-of-labels kernels (crf_general.hip):
windowed marginals, whole-contig marginals and Viterbi for CRFsuite models with L != 2
labels, and the same kernels forced onto 2-label models as an on-device cross-check of the
specialised ones.  Checker: the CPU oracle ([EXT] CRFsuite semantics) + brute-force
enumeration of label paths.  Tolerance: 1e-12 on marginals (north star: 1e-6), labels exact."""
import numpy as np
import pytest

from tests.helpers import golden_csr, synth_contigs, synth_model

pytestmark = pytest.mark.gpu

LABEL_COUNTS = [1, 3, 4, 5, 8, 13, 16, 17, 32]
LENGTHS = [1, 2, 3, 4, 7, 19, 20, 21, 40, 63, 64, 65, 200, 1500]


@pytest.fixture(scope="module")
def nat():
    from gecco_amd import _native

    assert _native.device_count() >= 1
    return _native


def _case(L, seed, A=300, extra=30):
    rng = np.random.default_rng(seed)
    w, trans = synth_model(A, rng, L=L)
    cptr, gptr, attr = synth_contigs(rng, LENGTHS + list(rng.integers(1, 120, size=extra)), A)
    return w, trans, cptr, gptr, attr


@pytest.mark.parametrize("L", LABEL_COUNTS)
def test_windowed_any_label_count(nat, L):
    from oracle import crf_oracle as orc

    w, trans, cptr, gptr, attr = _case(L, 100 + L)
    model = nat.Model.from_tables(w, trans)
    assert model.num_labels == L
    for W, step, pad, label in [(20, 1, True, L - 1), (5, 1, True, 0), (20, 3, True, L // 2), (20, 1, False, 0), (48, 7, True, 0)]:
        got = model.windowed_marginals(cptr, gptr, attr, W, step, label, pad)
        exp = orc.windowed_marginals(w, trans, cptr, gptr, attr, W, step, label, pad)
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.abs(got[ok] - exp[ok]).max() <= 1e-12, (L, W, step, pad, label)


@pytest.mark.parametrize("L", LABEL_COUNTS)
def test_full_marginals_any_label_count(nat, L):
    from oracle import crf_oracle as orc

    w, trans, cptr, gptr, attr = _case(L, 200 + L)
    model = nat.Model.from_tables(w, trans)
    marg, ln = model.marginals_full(cptr, gptr, attr)
    emarg, eln = orc.full_marginals(w, trans, cptr, gptr, attr)
    assert marg.shape == emarg.shape == (int(cptr[-1]), L)
    assert np.abs(marg - emarg).max() <= 1e-12
    np.testing.assert_allclose(marg.sum(axis=1), 1.0, atol=1e-13)
    assert np.abs(ln - eln).max() <= 1e-10 * max(1.0, np.abs(eln).max())


@pytest.mark.parametrize("L", LABEL_COUNTS)
def test_viterbi_any_label_count(nat, L):
    from oracle import crf_oracle as orc

    w, trans, cptr, gptr, attr = _case(L, 300 + L)
    model = nat.Model.from_tables(w, trans)
    y, sc = model.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y.astype(np.int32), ey)
    assert np.abs(sc - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())


@pytest.mark.parametrize("L", [3, 8, 16])
def test_empty_contigs_any_label_count(nat, L):
    """Contigs without genes have no chunk in the chunked whole-contig kernels: their log-partition and path score are 0
    (as the oracle's), not whatever an earlier batch left in the driver's buffers."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(900 + L)
    w, trans = synth_model(200, rng, L=L)
    model = nat.Model.from_tables(w, trans)
    # a first batch without empty contigs fills the per-contig buffers with non-zero values
    c0, g0, a0 = synth_contigs(rng, [30, 5, 70, 12, 1, 9, 200], 200)
    model.marginals_full(c0, g0, a0)
    model.viterbi(c0, g0, a0)
    cptr, gptr, attr = synth_contigs(rng, [0, 25, 0, 0, 3, 64, 0, 1, 0], 200)
    marg, ln = model.marginals_full(cptr, gptr, attr)
    emarg, eln = orc.full_marginals(w, trans, cptr, gptr, attr)
    assert np.abs(marg - emarg).max() <= 1e-12
    assert np.abs(ln - eln).max() <= 1e-10 * max(1.0, np.abs(eln).max())
    assert all(ln[c] == 0.0 for c in (0, 2, 3, 6, 8))
    y, sc = model.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y.astype(np.int32), ey)
    assert np.abs(sc - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())
    assert all(sc[c] == 0.0 for c in (0, 2, 3, 6, 8))


@pytest.mark.parametrize("L", [3, 5])
def test_viterbi_ties_first_argmax(nat, L):
    """Small-integer weights make exact ties common: the strict `<` update / first arg max of
    [EXT] crf1dc_viterbi must be reproduced label for label."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(17 + L)
    A = 40
    w = rng.integers(-2, 3, size=(A, L)).astype(np.float64)
    trans = rng.integers(-1, 2, size=(L, L)).astype(np.float64)
    cptr, gptr, attr = synth_contigs(rng, [1, 2, 5, 33, 64, 65, 300, 999], A)
    y, sc = nat.Model.from_tables(w, trans).viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y.astype(np.int32), ey)
    assert np.array_equal(sc, esc)


@pytest.mark.parametrize("L", [3, 4, 5, 6, 7, 8])
def test_lane_per_window_kernel(nat, L, monkeypatch):
    """3 to 8 labels take `gl_windowed_small` (one lane per window start, un-normalised recurrences, DPP maximum over
    the covering windows): every label, windows from 1 to 32 genes, steps, unpadded short contigs (skipped: irregular
    tiles), against the oracle and against the lane-group kernel (GECCO_CRF_GENERAL_GROUPS=1)."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(700 + L)
    A = 200
    w, trans = synth_model(A, rng, L=L)
    lengths = LENGTHS + list(rng.integers(1, 60, size=40)) + [237, 238, 474, 475, 3000] + list(rng.integers(1, 400, size=30))
    cptr, gptr, attr = synth_contigs(rng, lengths, A)
    model = nat.Model.from_tables(w, trans)
    assert nat.Plan(model, cptr, 20, 1, True, device=0).kernel_name == "gl_windowed_small"
    wide = [(21, 1, True, 2), (32, 1, True, L - 1), (32, 5, False, 2)] if L <= 4 else [(19, 1, True, 2), (20, 3, False, L - 1)]
    cases = [(20, 1, True, lab) for lab in range(L)] + [(1, 1, True, 0), (2, 1, True, 1), (20, 7, True, 1), (20, 1, False, 0),
                                                         (5, 4, False, 1)] + wide
    for W, step, pad, label in cases:
        got = model.windowed_marginals(cptr, gptr, attr, W, step, label, pad)
        exp = orc.windowed_marginals(w, trans, cptr, gptr, attr, W, step, label, pad)
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.abs(got[ok] - exp[ok]).max() <= 1e-12, (L, W, step, pad, label)
        monkeypatch.setenv("GECCO_CRF_GENERAL_GROUPS", "1")
        other = model.windowed_marginals(cptr, gptr, attr, W, step, label, pad)
        monkeypatch.delenv("GECCO_CRF_GENERAL_GROUPS")
        assert np.abs(got[ok] - other[ok]).max() <= 1e-12, (L, W, step, pad, label)


@pytest.mark.parametrize("L", [9, 12, 13, 16, 17, 20, 24, 25, 29, 32])
def test_matrix_core_window_kernel(nat, L, monkeypatch):
    """9 to 32 labels take `gl_windowed_mfma`: sixteen windows per wave as v_mfma_f64_16x16x4_f64 products (labels x
    windows), alpha / beta in the result registers from step to step, the maximum over the covering windows by LDS atomics.
    Every number of K-slices (3 to 8) and both tile counts, every queried label at L = 13, windows from 1 to 32 genes,
    steps, unpadded short contigs (skipped: irregular tiles), against the oracle (1e-12: the summation order of a step
    differs from CRFsuite's) and against the lane-group kernel (GECCO_CRF_GENERAL_GROUPS=1)."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(1700 + L)
    A = 200
    w, trans = synth_model(A, rng, L=L)
    lengths = LENGTHS + list(rng.integers(1, 60, size=40)) + [237, 238, 474, 475, 3000] + list(rng.integers(1, 400, size=30))
    cptr, gptr, attr = synth_contigs(rng, lengths, A)
    model = nat.Model.from_tables(w, trans)
    assert nat.Plan(model, cptr, 20, 1, True, device=0).kernel_name == "gl_windowed_mfma"
    labels = range(L) if L == 13 else (0, L // 2, L - 1)
    cases = [(20, 1, True, lab) for lab in labels] + [(1, 1, True, 0), (2, 1, True, 1), (20, 7, True, 1), (20, 1, False, 0),
                                                       (5, 4, False, 1), (21, 1, True, 2), (32, 1, True, L - 1), (32, 5, False, 2)]
    for W, step, pad, label in cases:
        got = model.windowed_marginals(cptr, gptr, attr, W, step, label, pad)
        exp = orc.windowed_marginals(w, trans, cptr, gptr, attr, W, step, label, pad)
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.abs(got[ok] - exp[ok]).max() <= 1e-12, (L, W, step, pad, label)
    monkeypatch.setenv("GECCO_CRF_GENERAL_GROUPS", "1")
    other = model.windowed_marginals(cptr, gptr, attr, 20, 1, 0, True)
    monkeypatch.delenv("GECCO_CRF_GENERAL_GROUPS")
    got = model.windowed_marginals(cptr, gptr, attr, 20, 1, 0, True)
    assert np.abs(got - other).max() <= 1e-12


def test_matrix_core_window_kernel_every_label_count(nat):
    """Every label count from 9 to 32 (every number of K-slices, labels that end inside a slice or a tile), W = 20 and a
    window of 32, the last label queried: gl_windowed_mfma against the oracle."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(4321)
    A = 120
    cptr, gptr, attr = synth_contigs(rng, [1, 19, 20, 21, 64, 237, 238, 500] + list(rng.integers(1, 90, size=25)), A)
    for L in range(9, 33):
        w, trans = synth_model(A, rng, L=L)
        model = nat.Model.from_tables(w, trans)
        for W, label in ((20, L - 1), (32, L // 3)):
            got = model.windowed_marginals(cptr, gptr, attr, W, 1, label, True)
            exp = orc.windowed_marginals(w, trans, cptr, gptr, attr, W, 1, label, True)
            assert np.abs(got - exp).max() <= 1e-12, (L, W, label)


def test_lane_per_window_kernel_range_guard(nat):
    """Its recurrences are un-normalised: transition weights whose spread times W - 1 stays under 600 keep every value
    in range (checked at the limit, with state weights as extreme as CRFsuite models get); beyond, the scaled kernel."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(77)
    A, L, W = 60, 3, 20
    w = np.clip(rng.laplace(0.0, 6.0, size=(A, L)), -40.0, 40.0)
    cptr, gptr, attr = synth_contigs(rng, [19, 20, 21, 300, 1000], A)
    for spread, kernel in ((31.5, "gl_windowed_small"), (31.6, "gl_windowed"), (80.0, "gl_windowed")):
        trans = rng.uniform(-1.0, 1.0, size=(L, L))
        trans[1, 2] = trans.max() - spread  # (W - 1) * 31.5 = 598.5
        trans[trans < trans[1, 2]] = trans[1, 2]
        model = nat.Model.from_tables(w, trans)
        assert nat.Plan(model, cptr, W, 1, True, device=0).kernel_name == kernel
        for label in range(L):
            got = model.windowed_marginals(cptr, gptr, attr, W, 1, label, True)
            exp = orc.windowed_marginals(w, trans, cptr, gptr, attr, W, 1, label, True)
            assert np.abs(got - exp).max() <= 1e-12, (spread, label)


@pytest.mark.parametrize("L", [3, 8])
def test_contig_sequential_kernels(nat, L, monkeypatch):
    """GECCO_CRF_GENERAL_CHUNKED=0: one group of lanes walks a whole contig (round 2's default for short contigs; kept as
    the strictly sequential form the chunked kernels are compared with)."""
    from oracle import crf_oracle as orc

    monkeypatch.setenv("GECCO_CRF_GENERAL_CHUNKED", "0")
    w, trans, cptr, gptr, attr = _case(L, 600 + L, extra=10)
    model = nat.Model.from_tables(w, trans)
    marg, ln = model.marginals_full(cptr, gptr, attr)
    emarg, eln = orc.full_marginals(w, trans, cptr, gptr, attr)
    assert np.abs(marg - emarg).max() <= 1e-12
    assert np.abs(ln - eln).max() <= 1e-10 * max(1.0, np.abs(eln).max())
    y, sc = model.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y.astype(np.int32), ey)
    assert np.abs(sc - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())


@pytest.mark.parametrize("L", [9, 13, 16, 17, 24, 32])
@pytest.mark.parametrize("mode", ["wave", "chunked", "split"])
def test_viterbi_wave_per_contig(nat, L, mode, monkeypatch):
    """9 to 32 labels: one wave per contig (gl_viterbi_wave: partial maxima merged in ascending source order, back-pointers as
    quads, scalar back-tracking), the chunked kernels, and the split (the longest contig chunked, the others by waves -- what
    the plan does with the long tail of a batch) give CRFsuite's labels and scores -- contigs of 1 .. 1 500 genes (every
    remainder of the quads and of the sixteen-quad rounds), and integer weights, where ties decide (strict `<`, first arg max)."""
    from oracle import crf_oracle as orc

    monkeypatch.setenv("GECCO_CRF_GENERAL_VITERBI", mode)
    w, trans, cptr, gptr, attr = _case(L, 700 + L, extra=60)
    model = nat.Model.from_tables(w, trans)
    y, sc = model.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y.astype(np.int32), ey)
    assert np.abs(sc - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())
    # ties: small integer weights (every sum exact), many equal path scores
    rng = np.random.default_rng(800 + L)
    wi = rng.integers(-2, 3, size=(40, L)).astype(np.float64)
    ti = rng.integers(-1, 2, size=(L, L)).astype(np.float64)
    c2, g2, a2 = synth_contigs(rng, [1, 2, 3, 4, 5, 8, 9, 63, 64, 65, 66, 127, 128, 129, 300] + [int(x) for x in rng.integers(1, 90, size=40)], 40)
    mi = nat.Model.from_tables(wi, ti)
    y2, s2 = mi.viterbi(c2, g2, a2)
    e2, es2 = orc.viterbi(wi, ti, c2, g2, a2)
    assert np.array_equal(y2.astype(np.int32), e2)
    assert np.array_equal(s2, es2)


@pytest.mark.parametrize("L", [9, 13, 16, 17, 24, 32])
@pytest.mark.parametrize("mode", ["wave", "chunked", "split"])
def test_marginals_wave_per_contig(nat, L, mode, monkeypatch):
    """Whole-contig marginals of 9 to 32 labels by a wave per contig (gl_marginals_wave), by the chunked kernels, and split (the
    longest contig chunked next to the waves of the others): 1e-12 against the oracle, rows summing to 1, log Z to 1e-10."""
    from oracle import crf_oracle as orc

    monkeypatch.setenv("GECCO_CRF_GENERAL_MARGINALS", mode)
    w, trans, cptr, gptr, attr = _case(L, 900 + L, extra=60)
    model = nat.Model.from_tables(w, trans)
    marg, ln = model.marginals_full(cptr, gptr, attr)
    emarg, eln = orc.full_marginals(w, trans, cptr, gptr, attr)
    assert marg.shape == emarg.shape
    assert np.abs(marg - emarg).max() <= 1e-12
    np.testing.assert_allclose(marg.sum(axis=1), 1.0, atol=1e-13)
    assert np.abs(ln - eln).max() <= 1e-10 * max(1.0, np.abs(eln).max())
    # empty contigs and one-gene contigs among the others
    rng = np.random.default_rng(950 + L)
    c2, g2, a2 = synth_contigs(rng, [0, 1, 0, 2, 17, 1, 300, 0], 300)
    m2, l2 = model.marginals_full(c2, g2, a2)
    em2, el2 = orc.full_marginals(w, trans, c2, g2, a2)
    assert np.abs(m2 - em2).max() <= 1e-12 and np.abs(l2 - el2).max() <= 1e-10 * max(1.0, np.abs(el2).max())


def test_viterbi_wave_is_chosen_for_batches_of_many_contigs(nat, monkeypatch):
    """Above 16 labels the plan splits a batch by its cost model: the long tail of the contig lengths through the chunked
    kernels, the rest through the wave-per-contig kernel (all of it, or none): every choice gives the oracle's labels; the
    choice only shows in the time."""
    from oracle import crf_oracle as orc

    monkeypatch.delenv("GECCO_CRF_GENERAL_VITERBI", raising=False)
    rng = np.random.default_rng(77)
    L = 20
    w, trans = synth_model(200, rng, L=L)
    model = nat.Model.from_tables(w, trans)
    for lengths in ([50] * 400, [3000, 20, 20], [60] * 300 + [900, 2500], [0, 0, 5, 0]):  # (all waves; chunked; split; empties)
        cptr, gptr, attr = synth_contigs(rng, lengths, 200)
        y, sc = model.viterbi(cptr, gptr, attr)
        ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
        assert np.array_equal(y.astype(np.int32), ey)
        assert np.abs(sc - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())


def test_three_labels_against_path_enumeration(nat):
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(5)
    L, A, T = 3, 12, 7
    w = rng.normal(0, 1.5, size=(A, L))
    trans = rng.normal(0, 1.0, size=(L, L))
    cptr, gptr, attr = synth_contigs(rng, [T], A)
    state = orc.state_scores(w, gptr, attr)
    model = nat.Model.from_tables(w, trans)
    marg, ln = model.marginals_full(cptr, gptr, attr)
    bm, bln = orc.brute_marginals(state, trans)
    assert np.abs(marg - bm).max() <= 1e-12 and abs(ln[0] - bln) <= 1e-10
    y, sc = model.viterbi(cptr, gptr, attr)
    by, bsc = orc.brute_viterbi(state, trans)
    assert np.array_equal(y.astype(np.int32), by) and abs(sc[0] - bsc) <= 1e-10
    # a window as long as the contig is the whole-contig marginal
    for label in range(L):
        p = model.windowed_marginals(cptr, gptr, attr, T, 1, label, True)
        assert np.abs(p - bm[:, label]).max() <= 1e-12


def test_two_label_model_through_general_kernels(nat, oracle_model, monkeypatch):
    """GECCO_CRF_FORCE_GENERAL=1: the shipped model on the any-L kernels must reproduce the golden
    table and agree with the specialised kernels."""
    import os

    from oracle import lcrf
    from tests.helpers import GOLDEN

    real = nat.Model.from_lcrf(lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))["blob"])
    ids, cptr, gptr, attr, expected, ann = golden_csr(oracle_model["attr_index"])
    rng = np.random.default_rng(9)
    rc, rg, ra = synth_contigs(rng, LENGTHS + list(rng.integers(1, 300, size=40)), oracle_model["state"].shape[0])
    fast = real.windowed_marginals(rc, rg, ra, 20, 1, 1, True)
    fast_full, fast_ln = real.marginals_full(rc, rg, ra)
    fast_y, fast_sc = real.viterbi(rc, rg, ra)
    monkeypatch.setenv("GECCO_CRF_FORCE_GENERAL", "1")
    p = real.windowed_marginals(cptr, gptr, attr, 20, 1, 1, True)
    assert np.abs(p - expected).max() <= 1e-12
    assert np.abs(real.windowed_marginals(rc, rg, ra, 20, 1, 1, True) - fast).max() <= 1e-12
    full, ln = real.marginals_full(rc, rg, ra)
    assert np.abs(full - fast_full).max() <= 1e-12 and np.abs(ln - fast_ln).max() <= 1e-9 * np.abs(ln).max()
    y, sc = real.viterbi(rc, rg, ra)
    assert np.array_equal(y, fast_y) and np.abs(sc - fast_sc).max() <= 1e-9 * np.abs(sc).max()


def test_window_too_long_is_reported(nat):
    rng = np.random.default_rng(3)
    w, trans = synth_model(20, rng, L=3)
    cptr, gptr, attr = synth_contigs(rng, [100], 20)
    with pytest.raises(nat.NativeError) as ei:
        nat.Model.from_tables(w, trans).windowed_marginals(cptr, gptr, attr, 49, 1, 0, True)
    assert ei.value.code == nat.EUNSUPPORTED


def test_too_many_labels_is_reported(nat):
    rng = np.random.default_rng(4)
    w, trans = synth_model(20, rng, L=33)
    cptr, gptr, attr = synth_contigs(rng, [30], 20)
    with pytest.raises(nat.NativeError) as ei:
        nat.Model.from_tables(w, trans).windowed_marginals(cptr, gptr, attr, 20, 1, 0, True)
    assert ei.value.code == nat.EUNSUPPORTED


# ---- long contigs: the chunked whole-contig kernels ("matrix-product scan for C5") ----
@pytest.mark.parametrize("L", [1, 2, 3, 5, 8, 17, 32])
def test_chunked_path_equals_oracle(nat, L, monkeypatch):
    """GECCO_CRF_GENERAL_CHUNKED=1 sends every batch through the chunked kernels (chunk matrices, vectors over
    chunks, replay inside chunks): same outputs as the oracle's sequential recursions, short contigs included."""
    from oracle import crf_oracle as orc

    monkeypatch.setenv("GECCO_CRF_GENERAL_CHUNKED", "1")
    monkeypatch.setenv("GECCO_CRF_FORCE_GENERAL", "1")
    w, trans, cptr, gptr, attr = _case(L, 400 + L, extra=10)
    model = nat.Model.from_tables(w, trans)
    marg, ln = model.marginals_full(cptr, gptr, attr)
    emarg, eln = orc.full_marginals(w, trans, cptr, gptr, attr)
    assert np.abs(marg - emarg).max() <= 1e-12
    assert np.abs(ln - eln).max() <= 1e-10 * max(1.0, np.abs(eln).max())
    y, sc = model.viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y.astype(np.int32), ey)
    assert np.abs(sc - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())


@pytest.mark.parametrize("L", [3, 5])
def test_chunked_viterbi_ties_first_argmax(nat, L, monkeypatch):
    from oracle import crf_oracle as orc

    monkeypatch.setenv("GECCO_CRF_GENERAL_CHUNKED", "1")
    rng = np.random.default_rng(27 + L)
    A = 40
    w = rng.integers(-2, 3, size=(A, L)).astype(np.float64)
    trans = rng.integers(-1, 2, size=(L, L)).astype(np.float64)
    cptr, gptr, attr = synth_contigs(rng, [1, 2, 63, 64, 65, 128, 129, 300, 5000], A)
    y, sc = nat.Model.from_tables(w, trans).viterbi(cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.array_equal(y.astype(np.int32), ey)
    assert np.array_equal(sc, esc)


@pytest.mark.parametrize("L", [3, 8])
def test_long_contig_any_label_count(nat, L):
    """A 50 000-gene contig (the C5 shape) next to short ones: taken by the chunked kernels on its own."""
    import time

    from oracle import crf_oracle as orc

    rng = np.random.default_rng(500 + L)
    A = 500
    w, trans = synth_model(A, rng, L=L)
    cptr, gptr, attr = synth_contigs(rng, [300, 50000, 7, 2049], A)
    model = nat.Model.from_tables(w, trans)
    t0 = time.perf_counter()
    marg, ln = model.marginals_full(cptr, gptr, attr)
    y, sc = model.viterbi(cptr, gptr, attr)
    dt = time.perf_counter() - t0
    emarg, eln = orc.full_marginals(w, trans, cptr, gptr, attr)
    ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
    assert np.abs(marg - emarg).max() <= 1e-12
    assert np.abs(ln - eln).max() <= 1e-10 * np.abs(eln).max()
    assert np.array_equal(y.astype(np.int32), ey)
    assert np.abs(sc - esc).max() <= 1e-9 * np.abs(esc).max()
    assert dt < 1.0  # the contig-sequential kernels need ~50 ms per pass for the long contig alone; this is a sanity bound


@pytest.mark.parametrize("L", [9, 11, 13, 15, 31])
def test_viterbi_split_wave_contig_next_to_a_chunked_one(nat, L, monkeypatch):
    """Split mode, odd label counts: wave contigs whose length is a multiple of four (whole quads of back-pointers) directly in
    front of the chunked contig, g0 * L not a multiple of four.  The wave kernel's quads hold rows 1 .. T - 1 from the first dword
    boundary of the contig's OWN T * L bytes (round 5 let them run up to 3 bytes into the next contig's row-0 bytes): a contig
    decoded by the chunked kernels on the side stream right behind a wave contig gets CRFsuite's labels, whatever the order."""
    from oracle import crf_oracle as orc

    monkeypatch.setenv("GECCO_CRF_GENERAL_VITERBI", "split")
    rng = np.random.default_rng(4100 + L)
    w, trans = synth_model(120, rng, L=L)
    model = nat.Model.from_tables(w, trans)
    for lengths in ([3, 8, 900, 12, 4], [1, 4, 4, 4, 1200, 4, 8], [5, 16, 64, 700, 20, 1], [2, 12, 1500]):
        cptr, gptr, attr = synth_contigs(rng, lengths, 120)
        for _ in range(3):  # (the two streams race differently from call to call)
            y, sc = model.viterbi(cptr, gptr, attr)
            ey, esc = orc.viterbi(w, trans, cptr, gptr, attr)
            assert np.array_equal(y.astype(np.int32), ey)
            assert np.abs(sc - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())
