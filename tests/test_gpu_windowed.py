"""GPU parity: HIP windowed-marginals path (through the C ABI) vs the CPU oracle and the
reference's golden table.  Tolerance: |dp| <= 1e-12 (north star asks 1e-6; fp64 throughout,
differences come from exp()/rounding order only)."""
import numpy as np
import pytest

from tests.helpers import golden_csr, synth_contigs, synth_model

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope="module")
def nat():
    from gecco_amd import _native

    assert _native.device_count() >= 1, "no HIP device: the GPU suite must run on an MI355X"
    return _native


@pytest.fixture(scope="module")
def real_model(nat, oracle_model):
    from oracle import lcrf
    import os
    from tests.helpers import GOLDEN

    st = lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))
    return nat.Model.from_lcrf(st["blob"])


def _cmp(got, exp):
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    err = np.abs(np.nan_to_num(got) - np.nan_to_num(exp)).max() if len(exp) else 0.0
    assert err <= TOL, err
    return err


def test_golden_fixture(real_model, oracle_model):
    ids, cptr, gptr, attr, expected, ann = golden_csr(oracle_model["attr_index"])
    p = real_model.windowed_marginals(cptr, gptr, attr, 20, 1, 1, True)
    assert np.abs(p - expected).max() <= 1e-14


@pytest.mark.parametrize("W,step,pad,label", [
    (20, 1, True, 1), (20, 1, False, 1), (20, 1, True, 0), (20, 3, True, 1), (20, 20, True, 1),
    (5, 1, True, 1), (5, 2, False, 1), (1, 1, True, 1), (2, 1, True, 1), (19, 1, True, 1),
    (21, 1, True, 1), (32, 1, True, 1), (32, 5, True, 0), (7, 7, True, 1),
])
def test_random_contigs_real_model(real_model, oracle_model, W, step, pad, label):
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(1000 * W + 10 * step + label)
    lengths = [1, 2, 3, W - 1 if W > 1 else 1, W, W + 1, 2 * W, 19, 20, 21, 39, 40, 41, 236, 237, 238, 255, 256, 257,
               274, 275, 276, 474, 475, 513, 1000] + list(rng.integers(1, 400, size=60))
    rng.shuffle(lengths)
    cptr, gptr, attr = synth_contigs(rng, lengths, oracle_model["state"].shape[0])
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, W, step, label, pad)
    got = real_model.windowed_marginals(cptr, gptr, attr, W, step, label, pad)
    _cmp(got, exp)


def test_c2_synthetic_model(nat):
    """configs[1]: 1k contigs x ~200 genes, A = 35k synthetic attributes."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(0x6ECC0)
    A = 35000
    w, trans = synth_model(A, rng)
    lengths = np.clip(np.round(rng.lognormal(np.log(200), 0.5, size=1000)), 5, 2000).astype(int)
    cptr, gptr, attr = synth_contigs(rng, lengths, A)
    model = nat.Model.from_tables(w, trans)
    got = model.windowed_marginals(cptr, gptr, attr, 20)
    exp = orc.windowed_marginals(w, trans, cptr, gptr, attr, 20)
    _cmp(got, exp)
    # cluster calls identical
    assert np.array_equal(got > 0.8, exp > 0.8)


def test_long_contig(real_model, oracle_model):
    """configs[4] shape at reduced size: one long contig spans many tiles."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(5)
    cptr, gptr, attr = synth_contigs(rng, [20000, 7], oracle_model["state"].shape[0])
    got = real_model.windowed_marginals(cptr, gptr, attr, 20)
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20)
    _cmp(got, exp)


def test_empty_and_degenerate(real_model):
    assert len(real_model.windowed_marginals([0], [0], [], 20)) == 0
    # genes without any domain: p is the prior path through empty items, identical for all genes
    p = real_model.windowed_marginals([0, 30], np.zeros(31, dtype=np.int32), [], 20)
    assert p.shape == (30,) and np.all(p > 0) and np.all(p < 1)
    # all contigs skipped (pad=False)
    p = real_model.windowed_marginals([0, 3, 5], [0, 1, 2, 3, 4, 5], [1, 2, 3, 4, 5], 20, pad=False)
    assert np.isnan(p).all()


def test_strong_transitions_need_rescale(nat):
    """Transition spread large enough that the kernel must renormalise mid-window."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(77)
    A = 300
    w, _ = synth_model(A, rng)
    trans = np.array([[20.0, -25.0], [-22.0, 18.0]])
    cptr, gptr, attr = synth_contigs(rng, [300, 45, 20, 8], A)
    model = nat.Model.from_tables(w, trans)
    for W in (20, 32, 9):
        got = model.windowed_marginals(cptr, gptr, attr, W)
        exp = orc.windowed_marginals(w, trans, cptr, gptr, attr, W)
        _cmp(got, exp)


def _logspace_windowed(w, trans, cptr, gptr, attr, W):
    """Independent log-domain windowed marginals (numpy logaddexp): immune to the exp()
    overflow that CRFsuite's (and hence the oracle's) probability-domain recursion hits."""
    out = np.zeros(cptr[-1])
    for c in range(len(cptr) - 1):
        g0, n = cptr[c], cptr[c + 1] - cptr[c]
        st = np.zeros((n, 2))
        for g in range(n):
            for a in attr[gptr[g0 + g]:gptr[g0 + g + 1]]:
                st[g] += w[a]
        assert n >= W
        prob = np.zeros(n)
        for s in range(n - W + 1):
            x = st[s:s + W]
            la = np.zeros((W, 2)); lb = np.zeros((W, 2))
            la[0] = x[0]
            for t in range(1, W):
                la[t] = x[t] + np.logaddexp(la[t - 1][0] + trans[0], la[t - 1][1] + trans[1])
            for t in range(W - 2, -1, -1):
                v = x[t + 1] + lb[t + 1]
                lb[t] = np.logaddexp(trans[:, 0] + v[0], trans[:, 1] + v[1])
            lz = np.logaddexp(la[-1][0], la[-1][1])
            prob[s:s + W] = np.maximum(prob[s:s + W], np.exp(la[:, 1] + lb[:, 1] - lz))
        out[g0:g0 + n] = prob
    return out


def test_extreme_state_scores(nat):
    """Saturated emissions: |s1-s0| of a few hundred must neither overflow nor NaN.  The
    probability-domain oracle is only usable while exp(s) is finite (sigma=40); beyond that
    (sigma=150) the HIP path is checked against a log-domain restatement."""
    from oracle import crf_oracle as orc

    trans = np.array([[2.67, -2.6], [-2.6, 2.57]])
    A = 50
    rng = np.random.default_rng(78)
    w = rng.normal(0, 40.0, size=(A, 2))
    cptr, gptr, attr = synth_contigs(rng, [200, 30], A)
    model = nat.Model.from_tables(w, trans)
    got = model.windowed_marginals(cptr, gptr, attr, 20)
    exp = orc.windowed_marginals(w, trans, cptr, gptr, attr, 20)
    assert np.isfinite(exp).all()
    _cmp(got, exp)

    w = rng.normal(0, 150.0, size=(A, 2))
    cptr, gptr, attr = synth_contigs(rng, [120, 25], A)
    model = nat.Model.from_tables(w, trans)
    got = model.windowed_marginals(cptr, gptr, attr, 20)
    assert np.isfinite(got).all() and (got >= 0).all() and (got <= 1).all()
    ref = _logspace_windowed(w, trans, cptr, gptr, attr, 20)
    assert np.abs(got - ref).max() <= 1e-9  # log-domain reference itself carries ~1e-13*|score| error


@pytest.mark.parametrize("W,step", [(33, 1), (50, 1), (64, 7), (100, 1)])
def test_large_windows_use_generic_kernel(nat, real_model, oracle_model, W, step):
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(W)
    cptr, gptr, attr = synth_contigs(rng, [5, W - 1, W, W + 1, 300, 77, 1200], oracle_model["state"].shape[0])
    plan = nat.Plan(real_model, cptr, W, step, True, device=-1)
    assert plan.kernel_name == "crf_windowed_generic_l2"
    got = real_model.windowed_marginals(cptr, gptr, attr, W, step)
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, W, step)
    _cmp(got, exp)


def test_huge_transition_spread_uses_generic_kernel(nat):
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(9)
    A = 100
    w, _ = synth_model(A, rng)
    trans = np.array([[100.0, -120.0], [-90.0, 80.0]])
    cptr, gptr, attr = synth_contigs(rng, [200, 20, 7], A)
    model = nat.Model.from_tables(w, trans)
    assert nat.Plan(model, cptr, 20, device=-1).kernel_name == "crf_windowed_generic_l2"
    got = model.windowed_marginals(cptr, gptr, attr, 20)
    exp = orc.windowed_marginals(w, trans, cptr, gptr, attr, 20)
    _cmp(got, exp)


def test_generic_kernel_cross_checks_fast_kernel(nat, real_model, oracle_model, monkeypatch):
    rng = np.random.default_rng(31)
    cptr, gptr, attr = synth_contigs(rng, list(rng.integers(1, 500, size=80)), oracle_model["state"].shape[0])
    fast = real_model.windowed_marginals(cptr, gptr, attr, 20, pad=False)
    monkeypatch.setenv("GECCO_CRF_FORCE_GENERIC", "1")
    slow = real_model.windowed_marginals(cptr, gptr, attr, 20, pad=False)
    _cmp(slow, fast)


@pytest.mark.parametrize("tiles", ["1", "3"])
def test_other_tiles_per_workgroup_geometries(nat, real_model, oracle_model, monkeypatch, tiles):
    """GECCO_CRF_TILES_PER_WG is an A/B switch of the plan geometry (default 2): every setting must
    give the same marginals, including padded / skipped contigs that make tiles irregular."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(int(tiles))
    cptr, gptr, attr = synth_contigs(rng, [3, 700, 19, 20, 21, 255, 256, 257, 1500] + list(rng.integers(1, 600, size=40)),
                                     oracle_model["state"].shape[0])
    monkeypatch.setenv("GECCO_CRF_TILES_PER_WG", tiles)
    for W, step, pad in [(20, 1, True), (20, 1, False), (7, 2, True), (32, 5, True)]:
        got = real_model.windowed_marginals(cptr, gptr, attr, W, step, 1, pad)
        exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, W, step, 1, pad)
        _cmp(got, exp)


def test_ratio_form_guard_at_its_limit(nat):
    """The fixed-window kernel takes the ratio form of the DP (one constant per slot, vectors growing
    like exp(sum of positive score differences)) only in workgroups where no window can exceed a sum of
    600: slots up to 200 / W = 10 add at most 200 per window, the differences of all other slots of the
    workgroup must add up to at most 400.  Runs just below either limit (a window at e^(198 + 389)),
    just above (max-normalised form), long saturated runs and strongly negative ones must all match the oracle."""
    from oracle import crf_oracle as orc
    from gecco_amd import synth

    w = np.zeros((7, 2))
    w[0] = (0.0, 29.9)    # d = +29.9: a leaning slot; 13 of them add up to 388.7 <= 400, 14 to 418.6
    w[1] = (0.0, 30.1)
    w[2] = (0.0, -50.0)
    w[3] = (700.0, 0.0)   # d = -700: exp underflows to 0
    w[4] = (1.0, 0.5)
    w[5] = (0.0, 9.9)     # d = +9.9: not leaning; twenty of them put 198 into a window
    w[6] = (500.0, -300.0)  # d = -800 (each exp() still finite for CRFsuite): the slot constant mu01 exp(d) flushes to zero -- with the start flag in its sign bit
    model = nat.Model.from_tables(w, synth.EMBEDDED_TRANS)
    rng = np.random.default_rng(1)
    runs = []
    for a, n in [(0, 600), (4, 300), (1, 600), (2, 50), (0, 40), (3, 30), (4, 700), (1, 25), (0, 500),
                 (4, 900), (5, 40), (0, 13), (5, 40), (4, 900), (5, 30), (0, 14), (5, 30), (4, 900), (0, 6), (5, 60), (0, 7), (4, 700),
                 (6, 30), (4, 100), (6, 1), (4, 50), (6, 2), (4, 300)]:
        runs += [a] * n
    runs = np.array(runs + list(rng.integers(0, 5, size=800)), dtype=np.int32)
    n = len(runs)
    gptr = np.arange(n + 1, dtype=np.int32)
    cptr = np.array([0, 900, 2500, 3645, n], dtype=np.int32)
    for label in (1, 0):
        got = model.windowed_marginals(cptr, gptr, runs, 20, 1, label, True)
        exp = orc.windowed_marginals(w, synth.EMBEDDED_TRANS, cptr, gptr, runs, 20, 1, label, True)
        assert np.all(np.isfinite(got))
        assert np.abs(got - exp).max() <= 1e-12
