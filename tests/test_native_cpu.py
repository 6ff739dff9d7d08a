"""CPU-side checks of the native boundary: the library loads, exports every symbol the header
declares, parses the shipped model exactly like the independent oracle parser, and fails
loudly (no CPU fallback) when asked to compute without a HIP device."""
import os
import re

import numpy as np
import pytest

from gecco_amd import _native as nat
from tests.helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def blob():
    from oracle import lcrf

    return lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))["blob"]


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gecco_crf.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gecco_crf_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 30
    lib = nat.load_library()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gecco_crf.h but not exported"
    assert declared == set(nat.SIGNATURES), declared ^ set(nat.SIGNATURES)
    assert lib.gecco_crf_version() >= 100


def test_model_tables_match_oracle_parser(blob, oracle_model):
    m = nat.Model.from_lcrf(blob)
    assert (m.num_labels, m.num_attrs, m.num_features) == (2, 2659, 4215)
    assert m.labels() == oracle_model["labels"] and m.attrs() == oracle_model["attrs"]
    w, present = m.state_weights()
    np.testing.assert_array_equal(w, oracle_model["state"])
    np.testing.assert_array_equal(present, oracle_model["state_mask"])
    t, tp = m.trans_weights()
    np.testing.assert_array_equal(t, oracle_model["trans"])
    assert tp.all()
    assert m.attr_id("PF00750") == 0 and m.attr_id("nope") == -1 and m.label_id("1") == 1
    ids = m.map_attrs(["PF00750", "zzz", "PF13471"])
    assert ids.tolist() == [0, -1, oracle_model["attr_index"]["PF13471"]]


@pytest.mark.parametrize("mutate", ["magic", "size", "version", "truncate", "feat", "cqdb"])
def test_malformed_models_are_rejected(blob, mutate):
    b = bytearray(blob)
    if mutate == "magic":
        b[0:4] = b"XXXX"
    elif mutate == "size":
        b[4:8] = (len(b) + 1).to_bytes(4, "little")
    elif mutate == "version":
        b[12:16] = (99).to_bytes(4, "little")
    elif mutate == "truncate":
        b = b[:1000]
    elif mutate == "feat":
        b[48:52] = b"TAEF"
    elif mutate == "cqdb":
        off = int.from_bytes(b[36:40], "little")
        b[off:off + 4] = b"BDQC"
    with pytest.raises(ValueError):
        nat.Model.from_lcrf(bytes(b))


def test_from_tables_roundtrip():
    rng = np.random.default_rng(0)
    w = rng.normal(size=(17, 3))
    t = rng.normal(size=(3, 3))
    m = nat.Model.from_tables(w, t)
    w2, _ = m.state_weights()
    t2, _ = m.trans_weights()
    np.testing.assert_array_equal(w, w2)
    np.testing.assert_array_equal(t, t2)
    assert m.labels() == ["0", "1", "2"] and m.attrs()[5] == "a5"


def test_argument_errors_mirror_reference_messages(blob):
    m = nat.Model.from_lcrf(blob)
    with pytest.raises(ValueError, match="Window size must be strictly positive"):
        m.windowed_marginals([0, 3], [0, 1, 2, 3], [1, 2, 3], 0)
    with pytest.raises(ValueError, match="Window step must be strictly positive and under `window_size`"):
        m.windowed_marginals([0, 3], [0, 1, 2, 3], [1, 2, 3], 5, step=6)
    with pytest.raises(ValueError, match="label out of range"):
        m.windowed_marginals([0, 3], [0, 1, 2, 3], [1, 2, 3], 5, label=2)


def test_host_only_plan_layout(blob):
    m = nat.Model.from_lcrf(blob)
    # contigs of 50, 10 (padded to 20) and 240 genes: windows = 31 + 1 + 221 (crf/__init__.py:239)
    p = nat.Plan(m, [0, 50, 60, 300], 20, 1, True, device=-1)
    assert (p.num_genes, p.num_windows) == (300, 253)
    # 310 slots (the 10-gene contig is padded to 20); a 256-lane workgroup runs two DP phases of
    # 256-(W-1) output slots each
    assert p.num_tiles == 1 and "crf_windowed" in p.kernel_name
    # ... for batches of up to 0.45 M slots ONE phase per workgroup (crf_plan.cpp: a batch that does not fill the chip), two beyond
    assert nat.Plan(m, [0, 5000], 20, 1, True, device=-1).num_tiles == -(-5000 // (256 - 19))
    assert nat.Plan(m, [0, 600000], 20, 1, True, device=-1).num_tiles == -(-600000 // (2 * (256 - 19)))
    p = nat.Plan(m, [0, 50, 60, 300], 20, 1, False, device=-1)
    assert p.num_windows == 252
    p = nat.Plan(m, [0], 20, device=-1)
    assert p.num_genes == 0 and p.num_tiles == 0
    with pytest.raises(nat.NativeError) as ei:
        nat.Plan(m, [0, 50], 20, device=-1).run_windowed(0, 0, 0)
    assert ei.value.code == nat.ENODEV


def test_host_only_plan_dispatch_by_label_count(blob, monkeypatch):
    rng = np.random.default_rng(1)
    m3 = nat.Model.from_tables(rng.normal(size=(9, 3)), rng.normal(size=(3, 3)))
    # 3 to 8 labels: one lane per window start, tiles of 256 - (W - 1) slots ...
    p3 = nat.Plan(m3, [0, 5000], 20, device=-1)
    assert p3.kernel_name == "gl_windowed_small" and p3.num_tiles == -(-5000 // (256 - 19))
    assert nat.Plan(m3, [0, 50], 33, device=-1).kernel_name == "gl_windowed"  # ... windows of up to 32 genes ...
    wide = rng.normal(size=(3, 3))
    wide[0, 1] = wide.min() - 40.0  # ... and transition weights W - 1 un-normalised steps cannot take out of the range
    assert nat.Plan(nat.Model.from_tables(rng.normal(size=(9, 3)), wide), [0, 50], 20, device=-1).kernel_name == "gl_windowed"
    m8 = nat.Model.from_tables(rng.normal(size=(9, 8)), rng.normal(size=(8, 8)))  # ... up to 8 labels at windows of up to 20
    assert nat.Plan(m8, [0, 50], 20, device=-1).kernel_name == "gl_windowed_small"
    assert nat.Plan(m8, [0, 50], 21, device=-1).kernel_name == "gl_windowed"
    # 9 to 32 labels: sixteen windows per wave on the fp64 matrix cores, same tiles, windows of up to 32 genes
    m9 = nat.Model.from_tables(rng.normal(size=(9, 9)), rng.normal(size=(9, 9)))
    p9 = nat.Plan(m9, [0, 5000], 20, device=-1)
    assert p9.kernel_name == "gl_windowed_mfma" and p9.num_tiles == -(-5000 // (256 - 19))
    m32 = nat.Model.from_tables(rng.normal(size=(9, 32)), rng.normal(size=(32, 32)))
    assert nat.Plan(m32, [0, 50], 32, device=-1).kernel_name == "gl_windowed_mfma"
    assert nat.Plan(m32, [0, 50], 33, device=-1).kernel_name == "gl_windowed"
    monkeypatch.setenv("GECCO_CRF_GENERAL_GROUPS", "1")
    assert nat.Plan(m3, [0, 50], 20, device=-1).kernel_name == "gl_windowed"
    monkeypatch.delenv("GECCO_CRF_GENERAL_GROUPS")
    m33 = nat.Model.from_tables(rng.normal(size=(9, 33)), rng.normal(size=(33, 33)))
    with pytest.raises(nat.NativeError) as ei:
        nat.Plan(m33, [0, 50], 20, device=-1)
    assert ei.value.code == nat.EUNSUPPORTED
    m2 = nat.Model.from_lcrf(blob)
    monkeypatch.setenv("GECCO_CRF_FORCE_GENERAL", "1")
    assert nat.Plan(m2, [0, 50], 20, device=-1).kernel_name == "gl_windowed"
    monkeypatch.delenv("GECCO_CRF_FORCE_GENERAL")
    assert "crf_windowed_l2" in nat.Plan(m2, [0, 50], 20, device=-1).kernel_name


@pytest.mark.skipif(nat.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback(blob):
    m = nat.Model.from_lcrf(blob)
    with pytest.raises(nat.NativeError) as ei:
        m.windowed_marginals([0, 30], np.arange(31), np.zeros(30, dtype=np.int32), 20)
    assert ei.value.code == nat.ENODEV


def test_pipelined_decode_kernel_keeps_its_window_tiles_out_of_scratch():
    """crf_decode_pipelined: the window tiles share a 64-VGPR kernel with the Viterbi workgroups, whose SGPR spills take
    one of the 64 registers; a tile value in scratch costs a quarter of the step (44 instead of 35 us, measured).  The
    cross-compiled assembly is checked: nothing but the entry reload of v0 touches scratch on the tile path."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_tile_path.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def test_exp_correctly_rounded_against_decimal():
    """The exp of reference-bits mode (gecco_crf_exp_correctly_rounded: the host build of the device's double-double code)
    returns the double nearest to exp(x): 12 000 random arguments over the range state scores take, and the edges."""
    import random
    from decimal import Decimal, getcontext

    from gecco_amd import _native

    getcontext().prec = 60
    random.seed(5)
    xs = [0.0, 1.0, -1.0, 700.0, -700.0, 709.7, -744.0, 1e-300, 12.65, -6.3] + [random.uniform(-60, 60) for _ in range(8000)] + \
         [random.uniform(-700, 700) for _ in range(2000)] + [random.gauss(0, 3) for _ in range(2000)]
    got = _native.exp_correctly_rounded(np.array(xs))
    want = np.array([float(Decimal(x).exp()) for x in xs])
    assert got.tobytes() == want.tobytes()
    assert np.isnan(_native.exp_correctly_rounded(np.array([np.nan]))[0])
    assert _native.exp_correctly_rounded(np.array([800.0]))[0] == np.inf and _native.exp_correctly_rounded(np.array([-800.0]))[0] == 0.0


def test_library_loads_the_wheels_hip_runtime_first():
    """A process that loads the library and imports torch afterwards must end up with ONE copy of libamdhip64 (the wheel's, when
    a torch is installed): loaded second, torch's copy would find no device.  GECCO_AMD_HIP_RUNTIME=system
    keeps the system's copy."""
    import importlib.util
    import subprocess
    import sys

    if importlib.util.find_spec("torch") is None:
        pytest.skip("no torch wheel here: the system's runtime is the only one")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from gecco_amd import _native\n_native.load_library()\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\nprint(libs)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    libs = eval(out.stdout.strip().splitlines()[-1])
    assert len(libs) == 1 and "/torch/lib/" in libs[0], libs
    env = dict(os.environ, GECCO_AMD_HIP_RUNTIME="system")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    libs = eval(out.stdout.strip().splitlines()[-1])
    assert len(libs) == 1 and "/torch/" not in libs[0], libs


def test_exp_fast_path_equals_the_slow_routine(tmp_path):
    """The correctly rounded exp takes a fast double-double path and accepts its result only when it is clear of a rounding
    boundary (Ziv): every accepted result must be the slow routine's, which is checked against 60-digit decimal arithmetic
    above.  4.4 M random and special arguments here (175 M when the fast path was written: no mismatch)."""
    import subprocess

    exe = str(tmp_path / "exp_fast_vs_slow")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "gecco_amd", "csrc"),
                           os.path.join(ROOT, "tests", "exp_fast_vs_slow.cpp"), "-o", exe])
    out = subprocess.run([exe, "2500000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout + out.stderr
