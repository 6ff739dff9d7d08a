"""Reference-bits mode (gecco_amd/csrc/crf_exact.hip): windowed marginals in CRFsuite's own operation order with a correctly
rounded exp.  Checked BIT FOR BIT against the oracle run with its own correctly rounded exp (libquadmath's expq -- an
independent implementation), and against the reference's output files: every one of the 46 + 37 + 2 probabilities the
BGC0001866 fixture prints comes out string-identical (/root/reference/galaxy/gecco.xml:83-111 compares whole files)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401  (before libgecco_crf.so: the wheel's own HIP runtime has to be the first one loaded)

from benchkit import latency  # noqa: E402
from tests.helpers import GOLDEN, golden_csr, read_tsv, synth_contigs, synth_model  # noqa: E402


@pytest.fixture(scope="module")
def nat():
    from gecco_amd import _native

    assert _native.device_count() >= 1, "no HIP device: the GPU suite must run on an MI355X"
    return _native


@pytest.fixture(scope="module")
def real_model(nat):
    return nat.Model.from_lcrf(latency.real_blob())


def _bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.tobytes() == b.tobytes()


def test_reference_bits_reproduce_the_fixture_strings(nat, real_model, oracle_model):
    """The 23 probabilities of BGC0001866.genes.tsv, as printed: repr() of the device's doubles == the file's strings."""
    ids, cptr, gptr, attr, exp, ann = golden_csr(oracle_model["attr_index"])
    genes = read_tsv(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    ses = nat.Session(real_model, [0])
    fast = ses.windowed_marginals(cptr, gptr, attr, 20)
    ses.set_reference_bits(True)
    p = ses.windowed_marginals(cptr, gptr, attr, 20)
    assert [repr(float(x)) for x in p] == [g["average_p"] for g in genes]
    assert float(np.abs(p - fast).max()) <= 1e-14  # (and the fast kernels are an ulp or a dozen away)
    ses.set_reference_bits(False)
    _bits(ses.windowed_marginals(cptr, gptr, attr, 20), fast)


@pytest.mark.parametrize("lengths,W,pad,step,direct", [
    ([50], 20, True, 1, True), ([7, 19, 20, 21, 0, 400, 1], 20, True, 1, True), ([7, 19, 20, 21, 0, 400, 1], 20, False, 1, False),
    ([300, 5, 60], 20, True, 3, True), ([2500, 30, 2049], 20, True, 1, False), ([120, 40, 9], 25, True, 1, True), ([90, 12], 32, False, 2, True),
    ([200] * 40, 20, True, 1, False), ([180] * 50 + [37], 20, True, 1, True),  # (the last: more tiles than exponentiate their own slots)
])
def test_reference_bits_equal_the_oracle_with_a_correctly_rounded_exp(nat, real_model, oracle_model, lengths, W, pad, step, direct):
    """Bit for bit, for padded / skipped / empty / long contigs, window sizes and steps other than GECCO's, through the direct
    path and through chunks; decode calls deliver the same p and CRFsuite's labels; cluster calls cut the same rows."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(len(lengths) * 100 + W)
    cptr, gptr, attr = synth_contigs(rng, lengths, real_model.num_attrs)
    with orc.correctly_rounded_exp():
        ep = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, W, step, 1, pad)
    ep_libm = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, W, step, 1, pad)
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    ses = nat.Session(real_model, [0])
    ses.set_reference_bits(True)
    if not direct:
        ses.set_chunk_genes(1024)
    p = ses.windowed_marginals(cptr, gptr, attr, W, step=step, pad=pad)
    assert ses.stats()["direct"] == (1 if direct or int(cptr[-1]) <= 1024 else 0)
    _bits(p, ep)
    # against libm's exp: the same bits on all but a handful of genes (where glibc's exp is not the correctly rounded one)
    differ = int(np.sum(~((p == ep_libm) | (np.isnan(p) & np.isnan(ep_libm)))))
    assert differ <= max(2, len(p) // 50), differ
    p2, y = ses.decode(cptr, gptr, attr, W, step=step, pad=pad)
    _bits(p2, ep)
    np.testing.assert_array_equal(y.astype(np.int32), ey)
    ann = (np.diff(gptr) > 0).astype(np.uint8)
    finite = np.sort(ep[~np.isnan(ep)])
    if len(finite) > 4:
        thr = float(0.5 * (finite[len(finite) // 2] + finite[len(finite) // 2 + 1]))
        seg, seg_p, seg_off, pp = ses.clusters(cptr, gptr, attr, ann, W, step=step, pad=pad, threshold=thr, n_cds=2, want_p=True)
        _bits(pp, ep)
        np.testing.assert_array_equal(seg, orc.segment(ep, ann, cptr, thr, 2, 0, True))
        for k, row in enumerate(seg):
            _bits(seg_p[seg_off[k]:seg_off[k + 1]], ep[row[2]:row[3]])


def test_reference_bits_on_a_synthetic_model(nat):
    """SURVEY.md 8d's weight law (windows that overflow the fast kernels' ratio form included): still the oracle's bits."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(8)
    w, trans = synth_model(3000, rng)
    model = nat.Model.from_tables(w, trans)
    cptr, gptr, attr = synth_contigs(rng, [150] * 30 + [17, 800], 3000)
    with orc.correctly_rounded_exp():
        ep = orc.windowed_marginals(w, trans, cptr, gptr, attr, 20, 1, 1, True)
    ses = nat.Session(model, [0])
    ses.set_reference_bits(True)
    _bits(ses.windowed_marginals(cptr, gptr, attr, 20), ep)
    # label 0 queried
    with orc.correctly_rounded_exp():
        e0 = orc.windowed_marginals(w, trans, cptr, gptr, attr, 20, 1, 0, True)
    _bits(ses.windowed_marginals(cptr, gptr, attr, 20, label=0), e0)


def test_reference_bits_unsupported_shapes(nat):
    rng = np.random.default_rng(9)
    w, trans = synth_model(50, rng, L=3)
    cptr, gptr, attr = synth_contigs(rng, [60], 50)
    ses = nat.Session(nat.Model.from_tables(w, trans), [0])
    ses.set_reference_bits(True)
    with pytest.raises(nat.NativeError):
        ses.windowed_marginals(cptr, gptr, attr, 20)
    w2, t2 = synth_model(50, rng)
    ses2 = nat.Session(nat.Model.from_tables(w2, t2), [0])
    ses2.set_reference_bits(True)
    with pytest.raises(nat.NativeError):
        ses2.windowed_marginals(cptr, gptr, attr, 40)


def test_cli_tables_are_the_reference_files_in_reference_bits_mode(tmp_path, capsys):
    """`python -m gecco_amd.predict --reference-bits` on the fixture: every cell of genes.tsv, features.tsv and of the CRF's
    columns of clusters.tsv is string-identical to the reference's file (`proteins` / `domains`: to the reference's current
    formula -- the fixture file predates it)."""
    from benchkit import levels

    res = levels.golden_table_identity(GOLDEN, str(tmp_path), reference_bits=True)
    with capsys.disabled():
        print("\n[table identity, reference bits]", {k: v for k, v in res.items() if k not in ("note", "mode")})
    for table, n_float in (("genes", 46), ("features", 37), ("clusters", 2)):
        t = res[table]
        assert t["rows"] == t["rows_expected"]
        assert t["exact_cells_differing"] == 0
        assert t["float_cells"] == n_float and t["float_cells_differing"] == 0
    assert res["clusters"]["formula_cells_differing"] == 0


def test_the_drop_in_class_answers_with_the_reference_bits_by_default(tmp_path, monkeypatch):
    """`ClusterCRF.reference_bits` None (the default): on whenever the mode covers the model -- GECCO's own model: yes; the
    CLI without a flag writes the reference's files; `--fast-kernels`, `reference_bits = False` and GECCO_AMD_REFERENCE_BITS=0
    turn it off; a window the mode does not cover falls back to the fast kernels instead of failing."""
    from gecco_amd import predict
    from gecco_amd.crf import ClusterCRF

    monkeypatch.delenv("GECCO_AMD_REFERENCE_BITS", raising=False)
    crf = ClusterCRF.trained(GOLDEN)
    assert crf.reference_bits is None and crf._reference_bits_now()
    _, cptr, gptr, attr, _, _ = golden_csr(crf.model._attr_index)
    p_default = crf.predict_probabilities_csr(cptr, gptr, attr)
    ses = crf._session()
    ses.set_reference_bits(True)
    _bits(p_default, ses.windowed_marginals(cptr, gptr, attr, 20))
    crf.reference_bits = False
    assert not crf._reference_bits_now()
    p_fast = crf.predict_probabilities_csr(cptr, gptr, attr)
    assert p_fast.tobytes() != p_default.tobytes() and np.max(np.abs(p_fast - p_default)) < 1e-14
    crf.reference_bits = None
    monkeypatch.setenv("GECCO_AMD_REFERENCE_BITS", "0")
    assert not crf._reference_bits_now()
    monkeypatch.delenv("GECCO_AMD_REFERENCE_BITS")
    crf.window_size = 40  # (not covered: W <= 32)
    assert not crf._reference_bits_now()
    assert np.isfinite(crf.predict_probabilities_csr(cptr, gptr, attr)).all()
    # the CLI without a flag: genes.tsv and features.tsv are the reference's files, byte for byte -- up to the line terminator:
    # the fixture files were written by the csv module ("\r\n"); the reference's current writer (polars write_csv,
    # /root/reference/gecco/model.py:770) and this one end lines with "\n"
    rc = predict.main(["--genes", os.path.join(GOLDEN, "BGC0001866.genes.tsv"), "--features",
                       os.path.join(GOLDEN, "BGC0001866.features.tsv"), "--model", GOLDEN, "-o", str(tmp_path)])
    assert rc == 0
    for table in ("genes", "features"):
        with open(tmp_path / f"BGC0001866.{table}.tsv", "rb") as fh:
            got = fh.read()
        with open(os.path.join(GOLDEN, f"BGC0001866.{table}.tsv"), "rb") as fh:
            ref = fh.read()
        assert b"\r" not in got and got == ref.replace(b"\r\n", b"\n")


def test_reference_bits_do_not_depend_on_how_the_batch_is_cut(nat):
    """A window's arithmetic is its own in reference-bits mode (no wave-level fallback, no tile-dependent summation): the
    probabilities of a 0.2 M-gene batch are bit-identical whether it goes through the direct path, 4 096-gene chunks, pieces of
    its longest contigs, two device entries or one 2^19-gene chunk -- and within 1e-13 of the fast kernels'."""
    from gecco_amd import synth

    wl = synth.workload("C2")
    model = nat.Model.from_tables(wl["w"], wl["trans"])
    cptr, gptr, attr = wl["contig_ptr"], wl["gene_ptr"], wl["attr_id"]
    # (one long contig in front, so that small chunks cut it into pieces)
    rng = np.random.default_rng(31)
    c2, g2, a2 = synth_contigs(rng, [30000], model.num_attrs)
    cptr = np.concatenate([c2, cptr[1:] + c2[-1]]).astype(np.int32)
    gptr = np.concatenate([g2, gptr[1:] + g2[-1]]).astype(np.int32)
    attr = np.concatenate([a2, attr]).astype(np.int32)
    ses = nat.Session(model, [0])
    fast = ses.windowed_marginals(cptr, gptr, attr, 20).copy()
    ses.set_reference_bits(True)
    base = ses.windowed_marginals(cptr, gptr, attr, 20).copy()
    assert float(np.abs(base - fast).max()) <= 1e-13
    for chunk, direct, entries in ((4096, 0, [0]), (1 << 16, 0, [0, 0]), (1 << 19, 1 << 20, [0]), (20000, 0, [0, 0, 0])):
        s2 = nat.Session(model, entries)
        s2.set_reference_bits(True)
        s2.set_chunk_genes(chunk)
        s2.set_direct_genes(direct)
        _bits(s2.windowed_marginals(cptr, gptr, attr, 20), base)
        p, y = s2.decode(cptr, gptr, attr, 20)
        _bits(p, base)


def test_device_override_keeps_the_mode(monkeypatch):
    """`device=` on the columnar entry points builds a session of its own: it must run in the object's mode (round 5: it
    silently ran the fast kernels), so the bits do not depend on whether `device` was passed."""
    from gecco_amd.crf import ClusterCRF

    monkeypatch.delenv("GECCO_AMD_REFERENCE_BITS", raising=False)
    crf = ClusterCRF.trained(GOLDEN)
    _, cptr, gptr, attr, _, _ = golden_csr(crf.model._attr_index)
    base = crf.predict_probabilities_csr(cptr, gptr, attr)
    crf.devices = [0, 0]  # (so that device=0 is "another device list" and takes the override path)
    _bits(crf.predict_probabilities_csr(cptr, gptr, attr, device=0), base)
    crf.reference_bits = False
    fast = crf.predict_probabilities_csr(cptr, gptr, attr, device=0)
    assert fast.tobytes() != base.tobytes() and np.max(np.abs(fast - base)) < 1e-14


def test_predict_tables_device_override_keeps_the_mode(monkeypatch):
    from gecco_amd import predict, tables
    from gecco_amd.crf import ClusterCRF

    monkeypatch.delenv("GECCO_AMD_REFERENCE_BITS", raising=False)
    crf = ClusterCRF.trained(GOLDEN)
    feats = tables.FeatureTable.load(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    genes = tables.GeneTable.load(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    a = predict.predict_tables(genes, feats, crf)
    b = predict.predict_tables(genes, feats, crf, device=0)
    _bits(np.asarray(a[0].average_p, dtype=np.float64), np.asarray(b[0].average_p, dtype=np.float64))
    crf.reference_bits = False
    c = predict.predict_tables(genes, feats, crf, device=0)
    assert np.asarray(c[0].average_p, dtype=np.float64).tobytes() != np.asarray(a[0].average_p, dtype=np.float64).tobytes()

_ENV_SWITCH_SCRIPT = r"""
import sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, {root!r})
from gecco_amd import _native as nat
from oracle import crf_oracle as orc
from tests.helpers import synth_contigs, synth_model

rng = np.random.default_rng(12)
w3, t3 = synth_model(50, rng, L=3)
c, g, a = synth_contigs(rng, [60, 5, 200], 50)
m3 = nat.Model.from_tables(w3, t3)
y, _ = m3.viterbi(c, g, a)
assert np.array_equal(y.astype(np.int32), orc.viterbi(w3, t3, c, g, a)[0])
m, _ = m3.marginals_full(c, g, a)
assert np.abs(m - orc.full_marginals(w3, t3, c, g, a)[0]).max() <= 1e-12
p3 = nat.Session(m3, [0]).windowed_marginals(c, g, a, 20)  # (another label count: the environment switch does not apply)
assert np.abs(p3 - orc.windowed_marginals(w3, t3, c, g, a, 20, 1, 1, True)).max() <= 1e-12
w2, t2 = synth_model(50, rng)
m2 = nat.Model.from_tables(w2, t2)
y2, _ = m2.viterbi(c, g, a)  # (a Viterbi-only layout of a 2-label model: its own kernels)
assert np.array_equal(y2.astype(np.int32), orc.viterbi(w2, t2, c, g, a)[0])
with orc.correctly_rounded_exp():
    e2 = orc.windowed_marginals(w2, t2, c, g, a, 20, 1, 1, True)
got = nat.Session(m2, [0]).windowed_marginals(c, g, a, 20)  # (windowed, 2 labels: the switch applies)
assert np.asarray(got).tobytes() == np.asarray(e2).tobytes()
p40 = nat.Session(m2, [0]).windowed_marginals(c, g, a, 40)  # (a window the mode does not cover: the fast kernels, no error)
assert np.abs(p40 - orc.windowed_marginals(w2, t2, c, g, a, 40, 1, 1, True)).max() <= 1e-12
print("ENV_SWITCH_OK")
"""


def test_environment_switch_leaves_viterbi_only_and_any_label_plans_alone():
    """GECCO_CRF_REFERENCE_BITS=1 is for windowed marginals of 2-label models: a Viterbi-only or whole-contig call, and a
    model with another label count, keep working (round 5: EUNSUPPORTED out of plan_build).  The library reads the variable
    once per process: a process of its own."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GECCO_CRF_REFERENCE_BITS="1")
    cp = subprocess.run([sys.executable, "-c", _ENV_SWITCH_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=300,
                        cwd=root)
    assert cp.returncode == 0 and "ENV_SWITCH_OK" in cp.stdout, cp.stdout[-2000:] + cp.stderr[-4000:]


def test_reference_bits_against_the_libm_oracle_on_c3(nat, capsys):
    """What the mode proves and what it does not.  CRFsuite calls the HOST'S libm `exp`; reference-bits mode computes CRFsuite's
    operation order with a CORRECTLY ROUNDED exp.  On C3 (2 M genes): bit-identical to the oracle run with a correctly rounded
    exp on every gene; against the oracle run with this box's glibc `exp` a small share of the genes differs (by a few ulps) --
    wherever glibc's result is not the correctly rounded one.  The fast kernels differ on most genes (a few ulps as well).
    The counts are printed and reported by bench.py (`parity.reference_bits_vs_libm_oracle`)."""
    from gecco_amd import synth
    from oracle import crf_oracle as orc

    wl = synth.workload("C3")
    model = nat.Model.from_tables(wl["w"], wl["trans"])
    cptr, gptr, attr = wl["contig_ptr"], wl["gene_ptr"], wl["attr_id"]
    libm = orc.windowed_marginals_mt(wl["w"], wl["trans"], cptr, gptr, attr, 20, 1, 1, True)
    with orc.correctly_rounded_exp():
        exact = orc.windowed_marginals_mt(wl["w"], wl["trans"], cptr, gptr, attr, 20, 1, 1, True)
    ses = nat.Session(model, [0])
    fast = np.asarray(ses.windowed_marginals(cptr, gptr, attr, 20)).copy()
    ses.set_reference_bits(True)
    bits = np.asarray(ses.windowed_marginals(cptr, gptr, attr, 20)).copy()

    def ulps(a, b):
        d = np.abs(a.view(np.int64) - b.view(np.int64))
        return int((d != 0).sum()), int(d.max())

    n = len(libm)
    rb_libm, rb_exact, fast_libm, oracle_modes = ulps(bits, libm), ulps(bits, exact), ulps(fast, libm), ulps(libm, exact)
    with capsys.disabled():
        print(f"\n[C3, {n} genes] reference-bits vs libm oracle: {rb_libm[0]} genes differ (max {rb_libm[1]} ulps); vs correctly rounded "
              f"oracle: {rb_exact[0]}; fast kernels vs libm oracle: {fast_libm[0]} (max {fast_libm[1]} ulps); the two oracles: {oracle_modes[0]}")
    assert rb_exact == (0, 0)
    assert rb_libm[0] == oracle_modes[0] and rb_libm[0] <= 0.02 * n and rb_libm[1] <= 64
    assert fast_libm[1] <= 4096 and float(np.abs(fast - libm).max()) <= 1e-12
    assert int(((bits > 0.8) != (libm > 0.8)).sum()) == 0 and int(((fast > 0.8) != (libm > 0.8)).sum()) == 0
