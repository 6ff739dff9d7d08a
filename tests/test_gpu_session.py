"""GPU parity of the batch driver (gecco_crf_session_*: chunked, pipelined, host buffers in / out) and
of the resident refiner epilogue, against the CPU oracle.  Chunk sizes are forced small so that one
batch goes through many chunks and every lane of the ring is reused."""
import numpy as np
import pytest

# PyTorch ships its own copy of the HIP runtime: it has to be the one this process loads first (importing torch
# after libgecco_crf.so has pulled in the system copy leaves torch without a device) -- INTEGRATION.md, section 3
import torch  # noqa: F401  (used by the device-array test below)

from tests.helpers import synth_contigs

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope="module")
def nat():
    from gecco_amd import _native

    assert _native.device_count() >= 1, "no HIP device: the GPU suite must run on an MI355X"
    return _native


@pytest.fixture(scope="module")
def real_model(nat):
    import os

    from oracle import lcrf
    from tests.helpers import GOLDEN

    st = lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))
    return nat.Model.from_lcrf(st["blob"])


def _batch(oracle_model, seed, n_contigs=300, hi=400):
    rng = np.random.default_rng(seed)
    lengths = list(rng.integers(1, hi, size=n_contigs)) + [1, 2, 19, 20, 21, 3000]
    rng.shuffle(lengths)
    return synth_contigs(rng, lengths, oracle_model["state"].shape[0])


def _threshold(p):
    """A cut near the median that no probability comes close to: device and oracle values differ in the
    last bits, so a gene sitting ON the threshold would be called differently."""
    v = np.sort(p[~np.isnan(p)])
    k = len(v) // 2
    while v[k + 1] - v[k] < 1e-6:
        k += 1
    return float(0.5 * (v[k] + v[k + 1]))


def _same(got, exp):
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    assert len(got) == 0 or np.abs(np.nan_to_num(got) - np.nan_to_num(exp)).max() <= TOL


@pytest.mark.parametrize("chunk,pad,pinned", [(1024, True, False), (5000, False, False), (20000, True, True), (1 << 19, True, False)])
def test_session_windowed_chunks(nat, real_model, oracle_model, chunk, pad, pinned):
    from oracle import crf_oracle as orc

    cptr, gptr, attr = _batch(oracle_model, 11)
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, pad)
    ses = nat.Session(real_model, [0])
    ses.set_chunk_genes(chunk)
    out = None
    if pinned:
        cptr, gptr, attr = nat.pinned_copy(cptr), nat.pinned_copy(gptr), nat.pinned_copy(attr)
        out = nat.pinned_empty(len(exp), np.float64)
    for _ in range(2):  # second pass: every buffer and plan of the ring is reused
        got = ses.windowed_marginals(cptr, gptr, attr, 20, 1, 1, pad, out=out)
        _same(got, exp)
    st = ses.stats()
    if chunk >= int(cptr[-1]):  # (a batch of one chunk, small enough for the direct path: no copy commands at all)
        assert st["direct"] == 1 and st["n_chunks"] == 1 and st["d2h_bytes"] == 0
        return
    assert st["direct"] == 0
    assert st["d2h_bytes"] == 8 * len(exp)
    if int(np.diff(cptr).max()) > 1.5 * chunk:
        # the 3 000-gene contig is cut into pieces of about a chunk, with a 19-gene halo either side (uploaded twice)
        assert st["n_chunks"] > (int(cptr[-1]) + chunk // 2) // chunk - 8
        assert st["h2d_bytes"] > 4 * (len(gptr) + len(attr))
        return
    assert st["n_chunks"] == max(1, min(len(cptr) - 1, (int(cptr[-1]) + chunk // 2) // chunk))
    assert st["h2d_bytes"] == 4 * (len(gptr) + st["n_chunks"] - 1 + len(attr))


def test_session_decode_and_one_shot_agree(nat, real_model, oracle_model):
    from oracle import crf_oracle as orc

    cptr, gptr, attr = _batch(oracle_model, 12)
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, True)
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    ses = nat.Session(real_model, [0])
    ses.set_chunk_genes(3000)
    p, y = ses.decode(cptr, gptr, attr, 20)
    _same(p, exp)
    np.testing.assert_array_equal(y.astype(np.int32), ey)
    # the one-shot entry points run on the model's own session: same numbers, bit for bit
    np.testing.assert_array_equal(real_model.windowed_marginals(cptr, gptr, attr, 20), p)
    y1, sc = real_model.viterbi(cptr, gptr, attr)
    np.testing.assert_array_equal(y1, y)
    _, esc = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.abs(sc - esc).max() <= 1e-9
    marg, ln = real_model.marginals_full(cptr, gptr, attr)
    em, eln = orc.full_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.abs(marg - em).max() <= TOL and np.abs(ln - eln).max() <= 1e-9


@pytest.mark.parametrize("devices,chunk", [([0], 1500), ([0, 0], 4000), ([0, 0, 0], 1 << 19)])
def test_session_decode_one_launch_per_chunk(nat, real_model, oracle_model, devices, chunk):
    """gecco_crf_session_decode drives every device with the pipelined launch: chunk k's window tiles and the Viterbi
    workgroups of the device's chunk k - 1 in one launch, a flush per device at the end.  Marginals and labels equal the
    oracle's, whatever the chunking and however the chunks are dealt to the device entries -- over batches with empty
    contigs, contigs shorter than the window, a contig longer than one scan block (that chunk takes separate launches
    behind the same call) and pad=False (skipped contigs: NaN marginals, labels all the same)."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(77 + len(devices))
    lengths = list(rng.integers(1, 400, size=250)) + [0, 0, 1, 2, 19, 20, 21, 0, 3000, 5, 2048, 2049, 7]
    rng.shuffle(lengths)
    cptr, gptr, attr = synth_contigs(rng, lengths, oracle_model["state"].shape[0])
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    ses = nat.Session(real_model, devices)
    ses.set_chunk_genes(chunk)
    for pad in (True, False):
        ep = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, pad)
        for _ in range(2):  # (twice: lanes and their plans are reused, pipelines start empty again)
            p, y = ses.decode(cptr, gptr, attr, 20, pad=pad)
            _same(p, ep)
            np.testing.assert_array_equal(y.astype(np.int32), ey)
        assert ses.stats()["n_chunks"] >= (3 if chunk < 100000 else 1)


@pytest.mark.parametrize("lengths", [[5000, 0, 0, 0, 0], [3000, 0, 0, 0, 0, 700, 0, 0, 0, 0, 0, 1500], [0, 0, 0, 0, 0, 0]])
def test_session_decode_empty_chunks_never_take_a_lane(nat, real_model, oracle_model, lengths):
    """Chunks without genes behind a scored one (contigs of length 0: one contig per chunk once the chunk size is small
    enough) must not come round to the lane whose labels are still pending: marginals, labels and path scores of the
    scored chunks stay the oracle's (advisor finding of round 4 on crf_session.cpp)."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(5)
    cptr, gptr, attr = synth_contigs(rng, lengths, oracle_model["state"].shape[0])
    ep = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, True)
    ey, es = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    ses = nat.Session(real_model, [0])
    ses.set_chunk_genes(1024)
    for _ in range(3):
        p, y = ses.decode(cptr, gptr, attr, 20)
        if sum(lengths):
            assert ses.stats()["n_chunks"] >= min(len(lengths), 5)
        _same(p, ep)
        np.testing.assert_array_equal(y.astype(np.int32), ey)
    y2, sc = real_model.viterbi(cptr, gptr, attr)
    np.testing.assert_array_equal(y2.astype(np.int32), ey)
    np.testing.assert_allclose(sc, es, rtol=1e-12, atol=1e-12)


def test_session_unknown_attribute_ids_carry_no_weight(nat, real_model, oracle_model):
    """ids outside the model's dictionary count as unknown attributes ([EXT] CRFsuite drops unknown names)."""
    from oracle import crf_oracle as orc

    A = oracle_model["state"].shape[0]
    cptr, gptr, attr = _batch(oracle_model, 13, n_contigs=40)
    rng = np.random.default_rng(0)
    bad = attr.copy()
    hit = rng.random(len(bad)) < 0.2
    bad[hit] = np.where(rng.random(int(hit.sum())) < 0.5, A + 5, -3)
    # the same batch with those attributes dropped
    keep = ~hit
    owner = np.repeat(np.arange(len(gptr) - 1), np.diff(gptr))
    g2 = np.concatenate([[0], np.cumsum(np.bincount(owner[keep], minlength=len(gptr) - 1))]).astype(np.int32)
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, g2, attr[keep], 20, 1, 1, True)
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, g2, attr[keep])
    ses = nat.Session(real_model, [0])
    _same(ses.windowed_marginals(cptr, gptr, bad, 20), exp)
    y, _ = real_model.viterbi(cptr, gptr, bad)
    np.testing.assert_array_equal(y.astype(np.int32), ey)


@pytest.mark.parametrize("chunk,pad,n_cds,edge,trim", [(2000, True, 3, 0, True), (1 << 19, False, 1, 2, True), (7000, False, 2, 0, False)])
def test_session_clusters_resident(nat, real_model, oracle_model, chunk, pad, n_cds, edge, trim):
    """predict_probabilities + refiner in one pass: rows equal the oracle's segmentation (one grouper per
    contig, like the CLI) of the oracle's probabilities, and only the rows' probabilities come back."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(14)
    cptr, gptr, attr = _batch(oracle_model, 14)
    ann = (np.diff(gptr) > 0).astype(np.uint8)
    ann[rng.random(len(ann)) < 0.05] ^= 1
    p_exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, pad)
    thr = _threshold(p_exp)  # the random batch has few genes above 0.8: cut where there are many runs
    exp = orc.segment(p_exp, ann, cptr, thr, n_cds, edge, trim, carry_state=False)
    assert len(exp) > 20
    ses = nat.Session(real_model, [0])
    ses.set_chunk_genes(chunk)
    seg, seg_p, seg_off, p = ses.clusters(cptr, gptr, attr, ann, 20, 1, 1, pad, thr, n_cds, edge, trim)
    assert p is None
    assert seg.tolist() == exp.tolist()
    for (c, num, a, b), o0, o1 in zip(seg.tolist(), seg_off[:-1], seg_off[1:]):
        assert o1 - o0 == b - a
        _same(seg_p[o0:o1], p_exp[a:b])
    # the rows arrive through pinned memory written by the kernels; only the probabilities of THEIR genes are downloaded
    assert ses.stats()["d2h_bytes"] == 8 * int(seg_off[-1]) < 8 * len(p_exp)
    ses.clusters(cptr, gptr, attr, ann, 20, 1, 1, pad, thr, n_cds, edge, trim, want_seg_p=False)
    assert ses.stats()["d2h_bytes"] == 0
    seg2, _, _, p2 = ses.clusters(cptr, gptr, attr, ann, 20, 1, 1, pad, thr, n_cds, edge, trim, want_p=True, want_seg_p=False)
    assert seg2.tolist() == exp.tolist()
    _same(p2, p_exp)


def test_plan_run_segment_on_device_arrays(nat, real_model, oracle_model):
    """gecco_crf_plan_run_segment chained behind gecco_crf_plan_run_windowed on one stream: p never leaves the device."""
    from oracle import crf_oracle as orc

    cptr, gptr, attr = _batch(oracle_model, 15)
    ann = (np.diff(gptr) > 0).astype(np.uint8)
    n = int(cptr[-1])
    dev = torch.device("cuda", 0)
    plan = nat.Plan(real_model, cptr, 20, 1, True, device=0)
    d_gp, d_at, d_ann = (torch.from_numpy(x).to(dev) for x in (gptr, attr, ann))
    d_p = torch.zeros(n, dtype=torch.float64, device=dev)
    d_seg = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    d_n = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    p_exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, True)
    thr = _threshold(p_exp)
    for carry in (False, True):
        plan.run_windowed(d_gp.data_ptr(), d_at.data_ptr(), d_p.data_ptr(), 1, stream)
        plan.run_segment(d_p.data_ptr(), d_ann.data_ptr(), d_seg.data_ptr(), n, d_n.data_ptr(), thr, 3, 0, True, carry, stream)
        torch.cuda.synchronize(dev)
        k = int(d_n.item())
        exp = orc.segment(p_exp, ann, cptr, thr, 3, 0, True, carry_state=carry)
        assert d_seg[:k].cpu().numpy().tolist() == exp.tolist()


def test_session_pinned_buffers_every_output(nat, real_model, oracle_model):
    """Pinned caller buffers (gecco_crf_host_alloc): every copy is asynchronous; same numbers as from pageable numpy
    arrays, bit for bit, also when chunks start at odd gene offsets."""
    cptr, gptr, attr = _batch(oracle_model, 17)
    pc, pg, pa = nat.pinned_copy(cptr), nat.pinned_copy(gptr), nat.pinned_copy(attr)
    n = int(cptr[-1])
    ses = nat.Session(real_model, [0])
    ses.set_chunk_genes(6001)
    p_z, y_z = ses.decode(pc, pg, pa, 20)
    out = nat.pinned_empty(n, np.float64)
    np.testing.assert_array_equal(ses.windowed_marginals(pc, pg, pa, 20, out=out), p_z)
    ann = (np.diff(gptr) > 0).astype(np.uint8)
    seg_z = ses.clusters(pc, pg, pa, ann, 20, threshold=_threshold(p_z), p_out=out)
    p_c, y_c = ses.decode(cptr, gptr, attr, 20)
    assert ses.stats()["h2d_bytes"] > 0
    np.testing.assert_array_equal(p_z, p_c)
    np.testing.assert_array_equal(y_z, y_c)
    seg_c = ses.clusters(cptr, gptr, attr, ann, 20, threshold=_threshold(p_z), p_out=np.empty(n))
    assert seg_z[0].tolist() == seg_c[0].tolist() and len(seg_z[0]) > 5
    np.testing.assert_array_equal(seg_z[3], seg_c[3])
    # the rows' probabilities into a caller buffer (pinned: the copy engine fills it at full rate)
    sp = nat.pinned_empty(n, np.float64)
    seg_b = ses.clusters(pc, pg, pa, ann, 20, threshold=_threshold(p_z), seg_p_out=sp)
    assert seg_b[0].tolist() == seg_c[0].tolist()
    np.testing.assert_array_equal(seg_b[1], seg_c[1])
    np.testing.assert_array_equal(seg_b[2], seg_c[2])


def test_session_argument_errors(nat, real_model):
    ses = nat.Session(real_model, [0])
    with pytest.raises(ValueError, match="Window size must be strictly positive"):
        ses.windowed_marginals([0, 1], [0, 0], [], 0)
    with pytest.raises(ValueError, match="Window step"):
        ses.windowed_marginals([0, 1], [0, 0], [], 5, step=6)
    with pytest.raises(ValueError, match="label out of range"):
        ses.windowed_marginals([0, 1], [0, 0], [], 5, label=2)
    with pytest.raises(ValueError, match="non-decreasing"):
        ses.windowed_marginals([0, 5, 3], [0] * 6, [], 5)
    assert len(ses.windowed_marginals([0], [0], [], 5)) == 0
    with pytest.raises(Exception):
        nat.Session(real_model, [nat.device_count()])


def test_session_several_device_entries(nat, real_model, oracle_model):
    """Chunks dealt longest-first over several device entries (the same physical device listed three times
    here: every entry has its own lanes and queue, exactly as three GPUs would)."""
    from oracle import crf_oracle as orc

    cptr, gptr, attr = _batch(oracle_model, 16, n_contigs=500)
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, True)
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    ses = nat.Session(real_model, [0, 0, 0])
    ses.set_chunk_genes(4096)
    p, y = ses.decode(cptr, gptr, attr, 20)
    _same(p, exp)
    np.testing.assert_array_equal(y.astype(np.int32), ey)
    ann = (np.diff(gptr) > 0).astype(np.uint8)
    thr = _threshold(exp)
    seg = ses.clusters(cptr, gptr, attr, ann, 20, threshold=thr)[0]
    assert seg.tolist() == orc.segment(exp, ann, cptr, thr, 3, 0, True, carry_state=False).tolist()


@pytest.mark.parametrize("chunk", [3000, 1 << 19])
def test_session_clusters_antismash(nat, real_model, oracle_model, chunk):
    """The batch driver with criterion "antismash": the genes' marker domains travel chunk by chunk with the CSR."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(16)
    cptr, gptr, attr = _batch(oracle_model, 16)
    n = int(cptr[-1])
    ann = (np.diff(gptr) > 0).astype(np.uint8)
    cnt = np.where(rng.random(n) < 0.4, rng.integers(1, 3, size=n), 0)
    mptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    mid = rng.integers(0, 20, size=int(mptr[-1])).astype(np.int32)
    p_exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, True)
    thr = _threshold(p_exp)
    exp = orc.segment_antismash(p_exp, ann, cptr, mptr, mid, thr, 3, 2, thr + 0.01, True)
    assert len(exp) > 10
    ses = nat.Session(real_model, [0])
    ses.set_chunk_genes(chunk)
    seg, seg_p, seg_off, _ = ses.clusters(cptr, gptr, attr, ann, 20, 1, 1, True, thr, 3, 0, True, criterion="antismash", n_biopfams=2,
                                          average_threshold=thr + 0.01, marker_ptr=mptr, marker_id=mid)
    assert seg.tolist() == exp.tolist()
    with pytest.raises(Exception, match="marker"):
        ses.clusters(cptr, gptr, attr, ann, 20, 1, 1, True, thr, 3, 0, True, criterion="antismash")


def test_degree_byte_wire_format_equals_row_pointers(nat, real_model, oracle_model):
    """gecco_crf_session_windowed_degrees: one degree byte per gene crosses PCIe instead of a 4-byte row pointer, the row
    pointers are rebuilt on the device (prefix sums, per chunk, with the caller's offsets).  Same bits as the row-pointer
    call over pageable and pinned buffers, several chunks, sizes around the scan's 4 096-gene blocks, genes with 0 and with
    many domains, an empty batch."""
    rng = np.random.default_rng(4242)
    A = oracle_model["state"].shape[0]
    ses = nat.Session(real_model, [0])
    for lengths, chunk in (([4095], 1 << 19), ([4096], 1 << 19), ([4097, 1], 1 << 19), (list(rng.integers(1, 500, size=400)), 4000),
                           (list(rng.integers(1, 3000, size=60)), 20000), ([], 1 << 19)):
        cptr, gptr, attr = synth_contigs(rng, lengths, A)
        ses.set_chunk_genes(chunk)
        deg = nat.degree_bytes(gptr)
        n = int(cptr[-1])
        a = ses.windowed_marginals(cptr, gptr, attr, 20).copy()
        b = ses.windowed_marginals(cptr, gptr, attr, 20, degree=deg).copy()
        assert np.array_equal(a, b, equal_nan=True), (lengths[:3], chunk)
        if n:
            cp, gp, at, dg = (nat.pinned_copy(x) for x in (cptr, gptr, attr, deg))
            out = nat.pinned_empty(n, np.float64)
            c = ses.windowed_marginals(cp, gp, at, 20, out=out, degree=dg)
            assert np.array_equal(a, c, equal_nan=True)
            # cluster calls take the same wire format (gecco_crf_session_clusters_degrees): same rows, same probabilities
            ann = (np.diff(gptr) > 0).astype(np.uint8)
            thr = _threshold(a) if n > 100 else 0.5
            r0 = ses.clusters(cptr, gptr, attr, ann, 20, threshold=thr)
            r1 = ses.clusters(cptr, gptr, attr, ann, 20, threshold=thr, degree=deg)
            assert r0[0].tolist() == r1[0].tolist() and np.array_equal(r0[1], r1[1]) and np.array_equal(r0[2], r1[2])
            # ... and `annotated` may be left to the degree bytes when it is "has a domain the model knows"
            r2 = ses.clusters(cptr, gptr, attr, None, 20, threshold=thr, degree=deg)
            assert r0[0].tolist() == r2[0].tolist() and np.array_equal(r0[1], r2[1])
            # ... and the attribute indices may cross as 16-bit words (gecco_crf_session_clusters_wire), pageable or pinned
            r3 = ses.clusters(cptr, gptr, attr.astype(np.uint16), None, 20, threshold=thr, degree=deg)
            r4 = ses.clusters(cp, gp, nat.pinned_copy(attr, np.uint16), None, 20, threshold=thr, degree=dg)
            for r in (r3, r4):
                assert r0[0].tolist() == r[0].tolist() and np.array_equal(r0[1], r[1]) and np.array_equal(r0[2], r[2])
            # the decode call takes both halves of the wire format too (gecco_crf_session_decode_wire): same bits, same labels
            p0, y0 = ses.decode(cptr, gptr, attr, 20)
            p0, y0 = p0.copy(), y0.copy()
            for a_, d_ in ((attr.astype(np.uint16), deg), (attr, deg), (attr.astype(np.uint16), None)):
                p1, y1 = ses.decode(cptr, gptr, a_, 20, degree=d_)
                assert np.array_equal(p0, p1, equal_nan=True) and np.array_equal(y0, y1)
            outy = nat.pinned_empty(n, np.int8)
            p2, y2 = ses.decode(cp, gp, nat.pinned_copy(attr, np.uint16), 20, out_p=out, out_y=outy, degree=dg)
            assert np.array_equal(p0, p2, equal_nan=True) and np.array_equal(y0, y2)
            p3, none = ses.decode(cp, gp, nat.pinned_copy(attr, np.uint16), 20, out_p=out, degree=dg, labels=False)
            assert none is None and np.array_equal(p0, p3, equal_nan=True)
    with pytest.raises(ValueError):
        nat.degree_bytes(np.array([0, 300], dtype=np.int32))
    # degree bytes that do not add up to the row pointers are refused (the device would read attributes out of bounds)
    cptr, gptr, attr = synth_contigs(rng, [300, 200], A)
    bad = nat.degree_bytes(gptr)
    bad[17] += 1
    with pytest.raises(ValueError, match="degree"):
        ses.windowed_marginals(cptr, gptr, attr, 20, degree=bad)
