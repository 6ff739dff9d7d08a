"""Randomised sweep of the batch driver: every case draws a batch shape (empty, one-gene, shorter-than-window, long contigs), a
window / step / padding / label, a chunk size, the direct path on or off, the compact wire format, pinned or pageable
buffers, reference-bits mode, one or two device entries -- and compares marginals, Viterbi labels and cluster rows with the CPU
oracle (1e-12; labels and rows exactly; reference bits: bit for bit).  The seeds are fixed: a failure names its case."""
import numpy as np
import pytest

import torch  # noqa: F401  (before libgecco_crf.so: the wheel's own HIP runtime has to be the first one loaded)

from tests.helpers import synth_contigs, synth_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from gecco_amd import _native

    assert _native.device_count() >= 1, "no HIP device: the GPU suite must run on an MI355X"
    return _native


def _lengths(rng):
    kind = int(rng.integers(0, 6))
    if kind == 0:  # a handful of short contigs, empty ones among them
        return [int(x) for x in rng.integers(0, 45, size=int(rng.integers(1, 12)))]
    if kind == 1:  # ONE contig
        return [int(rng.integers(1, 700))]
    if kind == 2:  # many small
        return [int(x) for x in rng.integers(0, 30, size=int(rng.integers(20, 120)))]
    if kind == 3:  # a long one among short ones (pieces, when the chunks are small)
        ls = [int(x) for x in rng.integers(1, 60, size=int(rng.integers(2, 10)))] + [int(rng.integers(1500, 5000))]
        rng.shuffle(ls)
        return ls
    if kind == 4:  # around the window size
        return [int(x) for x in rng.integers(15, 26, size=int(rng.integers(3, 30)))]
    return [int(x) for x in rng.integers(1, 400, size=int(rng.integers(5, 40)))]


def _pin(nat, a, pinned):
    return nat.pinned_copy(a) if pinned and a.size else a


@pytest.mark.parametrize("seed", range(300))
def test_session_against_the_oracle_on_a_random_case(nat, seed):
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(1000 + seed)
    A = int(rng.choice([40, 300, 3000]))
    w, trans = synth_model(A, rng)
    model = nat.Model.from_tables(w, trans)
    lengths = _lengths(rng)
    cptr, gptr, attr = synth_contigs(rng, lengths, A)
    n = int(cptr[-1])
    W = int(rng.choice([20, 20, 20, 5, 2, 32, 13]))
    step = int(rng.integers(1, W + 1)) if rng.random() < 0.3 else 1
    pad = bool(rng.random() < 0.7)
    label = int(rng.integers(0, 2))
    entries = [0, 0] if rng.random() < 0.25 else [0]
    ses = nat.Session(model, entries)
    chunk = int(rng.choice([256, 1024, 4096, 1 << 19]))
    ses.set_chunk_genes(chunk)
    if rng.random() < 0.3:
        ses.set_direct_genes(0)
    reference = bool(rng.random() < 0.35) and W <= 32
    ses.set_reference_bits(reference)
    pinned = bool(rng.random() < 0.5)
    wire = bool(rng.random() < 0.4)
    case = dict(seed=seed, lengths=lengths[:12], n=n, W=W, step=step, pad=pad, label=label, entries=entries, chunk=chunk,
                reference=reference, pinned=pinned, wire=wire)

    if reference:
        with orc.correctly_rounded_exp():
            ep = orc.windowed_marginals(w, trans, cptr, gptr, attr, W, step, label, pad)
    else:
        ep = orc.windowed_marginals(w, trans, cptr, gptr, attr, W, step, label, pad)
    ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)

    def same_p(got, what):
        assert got.shape == ep.shape, (what, case)
        assert np.array_equal(np.isnan(got), np.isnan(ep)), (what, case)
        if reference:
            assert got.tobytes() == ep.tobytes(), (what, case)
        elif n:
            assert np.abs(np.nan_to_num(got) - np.nan_to_num(ep)).max() <= 1e-12, (what, case)

    cp, gp = _pin(nat, cptr, pinned), _pin(nat, gptr, pinned)
    at = _pin(nat, attr, pinned)
    deg = _pin(nat, nat.degree_bytes(gptr), pinned) if wire else None
    at16 = _pin(nat, attr.astype(np.uint16), pinned) if wire else None
    out = nat.pinned_empty(max(n, 1), np.float64) if pinned else None

    p = ses.windowed_marginals(cp, gp, at, W, step=step, label=label, pad=pad, out=out, degree=deg)
    same_p(np.array(p[:n]), "windowed")
    p2, y = ses.decode(cp, gp, at16 if wire else at, W, step=step, label=label, pad=pad, degree=deg)
    same_p(p2, "decode p")
    assert np.array_equal(y.astype(np.int32), ey), ("decode labels", case)

    finite = np.sort(ep[~np.isnan(ep)])
    if len(finite) > 8:
        k = len(finite) // 2
        while k + 1 < len(finite) - 1 and finite[k + 1] - finite[k] < 1e-9:
            k += 1
        thr = float(0.5 * (finite[k] + finite[k + 1]))
        if finite[k + 1] - finite[k] >= 1e-9:
            ann = (np.diff(gptr) > 0).astype(np.uint8)
            n_cds = int(rng.integers(1, 4))
            trim = bool(rng.random() < 0.7)
            edge = int(rng.integers(0, 3))
            seg, seg_p, seg_off, pp = ses.clusters(cp, gp, at16 if wire else at, None if wire else ann, W, step=step, label=label, pad=pad,
                                                   threshold=thr, n_cds=n_cds, edge_distance=edge, trim=trim, want_p=True, degree=deg)
            same_p(pp, "clusters p")
            exp_seg = orc.segment(ep, ann, cptr, thr, n_cds, edge, trim)
            assert np.array_equal(seg, exp_seg), ("cluster rows", case)
            for i, row in enumerate(seg):
                got = seg_p[seg_off[i]:seg_off[i + 1]]
                assert got.shape[0] == row[3] - row[2], ("row probabilities", case)
                if reference:
                    assert got.tobytes() == ep[row[2]:row[3]].tobytes(), ("row probabilities", case)
                else:
                    assert np.abs(got - ep[row[2]:row[3]]).max() <= 1e-12, ("row probabilities", case)
