"""The batch driver's DIRECT path (small batches: one chunk, kernels on pinned host memory, no copy commands) against the
oracle and against the chunked path, bit for bit -- starting with BASELINE.json configs[0] ("C1": one 50-gene contig,
pretrained weights; /root/reference/tests/test_cli/test_run.py:35-70 -> gecco/crf/__init__.py:244-258)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401  (before libgecco_crf.so: the wheel's own HIP runtime has to be the first one loaded)

from benchkit import latency  # noqa: E402
from tests.helpers import synth_contigs, synth_model  # noqa: E402


@pytest.fixture(scope="module")
def nat():
    from gecco_amd import _native

    assert _native.device_count() >= 1, "no HIP device: the GPU suite must run on an MI355X"
    return _native


@pytest.fixture(scope="module")
def real_model(nat):
    return nat.Model.from_lcrf(latency.real_blob())


def _same(got, exp, tol=1e-12):
    got, exp = np.asarray(got), np.asarray(exp)
    assert got.shape == exp.shape
    nan = np.isnan(exp)
    assert np.array_equal(np.isnan(got), nan)
    if (~nan).any():
        assert float(np.abs(got[~nan] - exp[~nan]).max()) <= tol


def _bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    assert a.tobytes() == b.tobytes()


def test_c1_single_50_gene_contig(nat, real_model, oracle_model):
    """C1 through every entry a caller has: the one-shot ABI call, the session (marginals, decode, cluster calls) and the
    drop-in class on Gene objects -- marginals within 1e-12 of the oracle, labels and cluster rows identical, and the call
    takes the direct path."""
    from oracle import crf_oracle as orc

    from gecco_amd.crf import ClusterCRF
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand

    cptr, gptr, attr = latency.c1_batch(50, real_model.num_attrs)
    assert len(cptr) == 2 and cptr[-1] == 50
    ep = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, True)
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    _same(real_model.windowed_marginals(cptr, gptr, attr, 20), ep)
    ses = nat.Session(real_model, [0])
    p = ses.windowed_marginals(cptr, gptr, attr, 20)
    st = ses.stats()
    assert st["direct"] == 1 and st["n_chunks"] == 1 and st["h2d_bytes"] == 0 and st["d2h_bytes"] == 0
    _same(p, ep)
    p2, y = ses.decode(cptr, gptr, attr, 20)
    _bits(p2, p)
    np.testing.assert_array_equal(y.astype(np.int32), ey)
    # compact wire format, pinned buffers
    deg = nat.pinned_copy(nat.degree_bytes(gptr))
    at16 = nat.pinned_copy(attr, np.uint16)
    p3, y3 = ses.decode(nat.pinned_copy(cptr), nat.pinned_copy(gptr), at16, 20, degree=deg, out_p=nat.pinned_empty(50, np.float64),
                        out_y=nat.pinned_empty(50, np.int8))
    _bits(p3, p)
    _bits(y3, y)
    ann = (np.diff(gptr) > 0).astype(np.uint8)
    srt = np.sort(ep)
    thr = float(0.5 * (srt[len(ep) // 2] + srt[len(ep) // 2 + 1]))  # (so that the refiner has runs to cut; between two values)
    assert np.abs(ep - thr).min() > 1e-9
    for n_cds in (1, 3):
        seg, seg_p, seg_off, pp = ses.clusters(cptr, gptr, attr, ann, 20, threshold=thr, n_cds=n_cds, want_p=True)
        assert ses.stats()["direct"] == 1
        _bits(pp, p)
        exp_seg = orc.segment(ep, ann, cptr, thr, n_cds, 0, True)
        np.testing.assert_array_equal(seg, exp_seg)
        for k, row in enumerate(seg):
            _bits(seg_p[seg_off[k]:seg_off[k + 1]], p[row[2]:row[3]])
    # Gene objects through the drop-in class (repeated domain names collapse there: compare with the de-duplicated batch)
    crf = ClusterCRF.trained(latency.golden_dir())
    attrs = crf.model.attributes_
    src = Source("contig_c1")
    genes = [Gene(src, 1000 * g, 1000 * g + 900, Strand.Coding,
                  Protein(f"c1_{g}", None, [Domain(attrs[a], 10 * j + 1, 10 * j + 9, "Pfam", 1e-10, 1e-12) for j, a in enumerate(attr[gptr[g]:gptr[g + 1]])]))
             for g in range(50)]
    out = crf.predict_probabilities(genes)
    g2, a2 = latency.dedup_csr(gptr, attr)
    ep2 = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, g2, a2, 20, 1, 1, True)
    _same(np.array([g._probability for g in out]), ep2)


def _kinds(nat, ses, cptr, gptr, attr, W, pad, step, rng):
    """every kind of call the direct path serves -> a dict of outputs"""
    out = {}
    out["p"] = ses.windowed_marginals(cptr, gptr, attr, W, step=step, pad=pad)
    out["dp"], out["dy"] = ses.decode(cptr, gptr, attr, W, step=step, pad=pad)
    n = int(cptr[-1])
    if n:
        deg = nat.degree_bytes(gptr)
        out["wp"], out["wy"] = ses.decode(cptr, gptr, attr.astype(np.uint16), W, step=step, pad=pad, degree=deg)
        ann = (np.diff(gptr) > 0).astype(np.uint8)
        ann[rng.random(n) < 0.1] ^= 1
        finite = out["p"][~np.isnan(out["p"])]
        thr = float(np.sort(finite)[len(finite) // 2]) if len(finite) else 0.5
        out["seg"], out["seg_p"], out["seg_off"], out["cp"] = ses.clusters(cptr, gptr, attr, ann, W, step=step, pad=pad, threshold=thr, n_cds=2,
                                                                           want_p=True)
        s2 = ses.clusters(cptr, gptr, attr.astype(np.uint16), None, W, step=step, pad=pad, threshold=thr, n_cds=1, want_p=False,
                          want_seg_p=False, degree=deg)
        out["seg2"] = s2[0]
        # antismash criterion with marker domains
        mk = (rng.random(len(attr)) < 0.3)
        owner = np.repeat(np.arange(n), np.diff(gptr))
        mptr = np.concatenate([[0], np.cumsum(np.bincount(owner[mk], minlength=n))]).astype(np.int32)
        mid = (attr[mk] % 7).astype(np.int32)
        s3 = ses.clusters(cptr, gptr, attr, ann, W, step=step, pad=pad, threshold=thr, n_cds=2, criterion="antismash", n_biopfams=1,
                          average_threshold=thr, marker_ptr=mptr, marker_id=mid)
        out["seg3"], out["seg3_p"] = s3[0], s3[1]
    return out


@pytest.mark.parametrize("lengths,W,pad,step", [
    ([50], 20, True, 1), ([7], 20, True, 1), ([7], 20, False, 1), ([19, 20, 21, 0, 3, 400, 1, 0], 20, True, 1),
    ([19, 20, 21, 0, 3, 400, 1, 0], 20, False, 1), ([300, 5, 60], 20, True, 3), ([2500, 30, 2048, 2049], 20, True, 1),
    ([120, 40, 9], 25, True, 1), ([120, 40, 9], 25, False, 2), ([90, 50, 12], 40, True, 1), ([200] * 60, 20, True, 1),
    ([0, 0], 20, True, 1),
])
def test_direct_equals_chunked_bits(nat, real_model, lengths, W, pad, step):
    """Same bits from the direct path and from the chunked path (copy commands, three streams) for every kind of call, over
    padded / skipped / empty / long contigs, the W != 20 kernels and the generic window kernel (atomic maxima: p stays in
    device memory there)."""
    rng = np.random.default_rng(len(lengths) * 1000 + W + step)
    cptr, gptr, attr = synth_contigs(rng, lengths, real_model.num_attrs)
    direct, chunked = nat.Session(real_model, [0]), nat.Session(real_model, [0])
    chunked.set_direct_genes(0)
    for _ in range(2):  # (twice: the lane's plan, staging block and workspaces are reused)
        a = _kinds(nat, direct, cptr, gptr, attr, W, pad, step, np.random.default_rng(1))
        assert direct.stats()["direct"] == (1 if cptr[-1] > 0 else 0)
        b = _kinds(nat, chunked, cptr, gptr, attr, W, pad, step, np.random.default_rng(1))
        assert chunked.stats()["direct"] == 0
        assert a.keys() == b.keys()
        for k in a:
            _bits(a[k], b[k])


def test_direct_on_pinned_caller_buffers(nat, real_model, oracle_model):
    """Outputs of 32 KB and more in pinned caller memory are written in place by the kernels (no staging copy) and inputs of
    64 KB and more go to device memory by a copy command on the compute stream: 20 000 genes."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(3)
    cptr, gptr, attr = synth_contigs(rng, [200] * 100, real_model.num_attrs)
    n = int(cptr[-1])
    ep = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, True)
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    ses = nat.Session(real_model, [0])
    cp, gp, at = nat.pinned_copy(cptr), nat.pinned_copy(gptr), nat.pinned_copy(attr)
    out_p, out_y = nat.pinned_empty(n, np.float64), nat.pinned_empty(n, np.int8)
    for _ in range(2):
        out_p[:] = -1.0
        out_y[:] = 9
        p, y = ses.decode(cp, gp, at, 20, out_p=out_p, out_y=out_y)
        assert ses.stats()["direct"] == 1
        _same(p, ep)
        np.testing.assert_array_equal(y.astype(np.int32), ey)
    # a slice of a larger pinned batch (offsets into the caller's arrays stay the caller's)
    out_p[:] = -1.0
    p = ses.windowed_marginals(cp, gp, at, 20, out=out_p)
    _same(p, ep)


def test_direct_any_label_count(nat):
    """A 3-label model: the any-L window kernels accumulate with atomic maxima, so p stays in device memory on the direct
    path too; same bits as the chunked path, oracle within 1e-12."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(11)
    w, trans = synth_model(300, rng, L=3)
    model = nat.Model.from_tables(w, trans)
    cptr, gptr, attr = synth_contigs(rng, [60, 5, 200, 33], 300)
    ep = orc.windowed_marginals(w, trans, cptr, gptr, attr, 20, 1, 1, True)
    ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)
    direct, chunked = nat.Session(model, [0]), nat.Session(model, [0])
    chunked.set_direct_genes(0)
    pa, ya = direct.decode(cptr, gptr, attr, 20)
    pb, yb = chunked.decode(cptr, gptr, attr, 20)
    assert direct.stats()["direct"] == 1 and chunked.stats()["direct"] == 0
    _bits(pa, pb)
    _bits(ya, yb)
    _same(pa, ep)
    np.testing.assert_array_equal(ya.astype(np.int32), ey)


def test_direct_integer_weights_exact_ties(nat):
    """Integer-valued weights: exact ties everywhere, the Viterbi decoder's exact pass (CRFsuite's own recursion on freshly
    summed state scores, read from host memory on this path) decides; labels == oracle."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(21)
    A = 40
    w = rng.integers(-2, 3, size=(A, 2)).astype(np.float64)
    trans = np.array([[1.0, -1.0], [-1.0, 1.0]])
    model = nat.Model.from_tables(w, trans)
    cptr, gptr, attr = synth_contigs(rng, [150, 40, 700, 9, 1], A)
    ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)
    ses = nat.Session(model, [0])
    for _ in range(2):
        _, y = ses.decode(cptr, gptr, attr, 20)
        assert ses.stats()["direct"] == 1
        np.testing.assert_array_equal(y.astype(np.int32), ey)


def test_direct_threshold_follows_chunk_size(nat, real_model):
    """A caller who asks for chunks smaller than the batch gets chunks; set_direct_genes(0) switches the path off."""
    rng = np.random.default_rng(2)
    cptr, gptr, attr = synth_contigs(rng, [300] * 20, real_model.num_attrs)
    ses = nat.Session(real_model, [0])
    a = ses.windowed_marginals(cptr, gptr, attr, 20)
    assert ses.stats()["direct"] == 1
    ses.set_chunk_genes(2000)
    b = ses.windowed_marginals(cptr, gptr, attr, 20)
    assert ses.stats()["direct"] == 0 and ses.stats()["n_chunks"] >= 3
    ses.set_chunk_genes(1 << 19)
    ses.set_direct_genes(1000)
    c = ses.windowed_marginals(cptr, gptr, attr, 20)
    assert ses.stats()["direct"] == 0 and ses.stats()["n_chunks"] == 1
    _bits(a, b)
    _bits(a, c)


def test_set_direct_genes_minus_one_restores_the_defaults(nat):
    """include/gecco_crf.h documents -1 as "the defaults" (round 5's C ABI refused it): an override can be undone."""
    rng = np.random.default_rng(3)
    w, trans = synth_model(60, rng)
    model = nat.Model.from_tables(w, trans)
    cptr, gptr, attr = synth_contigs(rng, [300, 40, 7], 60)
    ses = nat.Session(model, [0])
    base = ses.windowed_marginals(cptr, gptr, attr, 20).copy()
    ses.set_direct_genes(0)
    assert np.abs(ses.windowed_marginals(cptr, gptr, attr, 20) - base).max() <= 1e-14
    ses.set_direct_genes(-1)
    assert np.array_equal(ses.windowed_marginals(cptr, gptr, attr, 20), base)
    with pytest.raises(ValueError):
        ses.set_direct_genes(-2)
