"""GPU: the resident / asynchronous half of the C ABI (gecco_crf_plan_*): caller-owned device
buffers (torch is only the allocator here), caller's stream, one plan reused across launches.
Results must equal the one-shot host entry points and the CPU oracle."""
import numpy as np
import pytest

from tests.helpers import synth_contigs

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def nat():
    from gecco_amd import _native

    assert _native.device_count() >= 1
    return _native


@pytest.fixture(scope="module")
def real(nat):
    import os

    from oracle import lcrf
    from tests.helpers import GOLDEN

    return nat.Model.from_lcrf(lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))["blob"])


def _batch(oracle_model, seed, lengths=None):
    rng = np.random.default_rng(seed)
    lengths = lengths if lengths is not None else [1, 5, 19, 20, 21, 64, 300, 1000] + list(rng.integers(1, 400, size=60))
    return synth_contigs(rng, lengths, oracle_model["state"].shape[0])


def _dev(*arrays):
    return [torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in arrays]


@pytest.mark.parametrize("label", [1, 0])
@pytest.mark.parametrize("pad", [True, False])
def test_decode_equals_separate_launches(nat, real, oracle_model, label, pad):
    from oracle import crf_oracle as orc

    cptr, gptr, attr = _batch(oracle_model, 40 + label)
    n = int(cptr[-1])
    d_gp, d_at = _dev(gptr, attr)
    plan = nat.Plan(real, cptr, 20, 1, pad, device=0)
    p1 = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    y1 = torch.zeros(n, dtype=torch.int8, device="cuda:0")
    s1 = torch.zeros(len(cptr) - 1, dtype=torch.float64, device="cuda:0")
    p2, y2, s2 = torch.zeros_like(p1), torch.zeros_like(y1), torch.zeros_like(s1)
    stream = torch.cuda.Stream(device="cuda:0")
    with torch.cuda.stream(stream):
        plan.run_windowed(d_gp.data_ptr(), d_at.data_ptr(), p1.data_ptr(), label, stream.cuda_stream)
        plan.run_viterbi(d_gp.data_ptr(), d_at.data_ptr(), y1.data_ptr(), s1.data_ptr(), stream.cuda_stream)
        for _ in range(2):  # a plan is reusable
            plan.run_decode(d_gp.data_ptr(), d_at.data_ptr(), p2.data_ptr(), y2.data_ptr(), label, s2.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    a, b = p1.cpu().numpy(), p2.cpu().numpy()
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    assert torch.equal(y1, y2) and torch.equal(s1, s2)
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, label, pad)
    ok = ~np.isnan(exp)
    assert np.array_equal(np.isnan(b), ~ok) and np.abs(b[ok] - exp[ok]).max() <= 1e-12
    ey, esc = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.array_equal(y2.cpu().numpy().astype(np.int32), ey)
    assert np.abs(s2.cpu().numpy() - esc).max() <= 1e-9 * max(1.0, np.abs(esc).max())
    # without path scores the decoders share 8 B/gene (score differences) instead of 16
    p3, y3 = torch.zeros_like(p1), torch.zeros_like(y1)
    plan.run_decode(d_gp.data_ptr(), d_at.data_ptr(), p3.data_ptr(), y3.data_ptr(), label, 0, 0)
    torch.cuda.synchronize()
    c = p3.cpu().numpy()
    assert np.array_equal(np.isnan(c), np.isnan(b)) and np.array_equal(c[~np.isnan(c)], b[~np.isnan(b)])
    assert torch.equal(y3, y2)


def test_resident_calls_equal_one_shot(nat, real, oracle_model):
    cptr, gptr, attr = _batch(oracle_model, 7)
    n = int(cptr[-1])
    d_gp, d_at = _dev(gptr, attr)
    plan = nat.Plan(real, cptr, 20, 1, True, device=0)
    p = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    marg = torch.zeros(n, 2, dtype=torch.float64, device="cuda:0")
    ln = torch.zeros(len(cptr) - 1, dtype=torch.float64, device="cuda:0")
    plan.run_windowed(d_gp.data_ptr(), d_at.data_ptr(), p.data_ptr(), 1)
    plan.run_marginals_full(d_gp.data_ptr(), d_at.data_ptr(), marg.data_ptr(), ln.data_ptr())
    ms = plan.time_windowed(d_gp.data_ptr(), d_at.data_ptr(), p.data_ptr(), 1, warmup=1, iters=3)
    torch.cuda.synchronize()
    assert ms > 0
    assert np.array_equal(p.cpu().numpy(), real.windowed_marginals(cptr, gptr, attr, 20))
    m1, l1 = real.marginals_full(cptr, gptr, attr)
    assert np.array_equal(marg.cpu().numpy(), m1) and np.array_equal(ln.cpu().numpy(), l1)


def test_batch_whose_attribute_array_ends_inside_a_gather(nat, real, oracle_model):
    """The windowed kernel loads 8 attribute ids per gene unconditionally through a bounds-checked
    buffer descriptor: the last genes' loads run past the end of attr_id and must read as absent."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(12)
    A = oracle_model["state"].shape[0]
    gptr = np.arange(0, 41, dtype=np.int32)  # 40 genes, one domain each, nothing after the last id
    attr = rng.integers(0, A, size=40).astype(np.int32)
    cptr = np.array([0, 40], dtype=np.int32)
    d_gp = torch.from_numpy(gptr).to("cuda:0")
    d_at = torch.from_numpy(attr).to("cuda:0")  # exact-size allocation
    p = torch.zeros(40, dtype=torch.float64, device="cuda:0")
    nat.Plan(real, cptr, 20, 1, True, device=0).run_windowed(d_gp.data_ptr(), d_at.data_ptr(), p.data_ptr(), 1)
    torch.cuda.synchronize()
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20)
    assert np.abs(p.cpu().numpy() - exp).max() <= 1e-12
    # genes with more than 8 domains take a second trip of the gather loop
    gptr2 = np.array([0, 11, 11, 30] + list(range(31, 58)), dtype=np.int32)
    attr2 = rng.integers(0, A, size=int(gptr2[-1])).astype(np.int32)
    cptr2 = np.array([0, len(gptr2) - 1], dtype=np.int32)
    got = real.windowed_marginals(cptr2, gptr2, attr2, 20)
    exp2 = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr2, gptr2, attr2, 20)
    assert np.abs(got - exp2).max() <= 1e-12


def test_decode_edge_batches(nat, real, oracle_model):
    """Empty batch, empty contigs in the middle, and a batch in which pad=False skips every contig
    (marginals are all 'no prediction', Viterbi labels are still decoded)."""
    from oracle import crf_oracle as orc

    plan = nat.Plan(real, [0], 20, 1, True, device=0)
    plan.run_decode(0, 0, 0, 0)  # nothing to do, nothing dereferenced
    plan.run_windowed(0, 0, 0)
    plan.run_viterbi(0, 0, 0)

    rng = np.random.default_rng(5)
    A = oracle_model["state"].shape[0]
    cptr, gptr, attr = synth_contigs(rng, [30, 0, 0, 25, 0, 400], A)
    d_gp, d_at = _dev(gptr, attr)
    n = int(cptr[-1])
    p = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    y = torch.zeros(n, dtype=torch.int8, device="cuda:0")
    nat.Plan(real, cptr, 20, 1, True, device=0).run_decode(d_gp.data_ptr(), d_at.data_ptr(), p.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20)
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.abs(p.cpu().numpy() - exp).max() <= 1e-12 and np.array_equal(y.cpu().numpy().astype(np.int32), ey)

    cptr, gptr, attr = synth_contigs(rng, [5, 19, 1, 7], A)
    d_gp, d_at = _dev(gptr, attr)
    n = int(cptr[-1])
    p = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    y = torch.full((n,), 7, dtype=torch.int8, device="cuda:0")
    nat.Plan(real, cptr, 20, 1, False, device=0).run_decode(d_gp.data_ptr(), d_at.data_ptr(), p.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    assert bool(torch.isnan(p).all())
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.array_equal(y.cpu().numpy().astype(np.int32), ey)


def test_pipelined_decode_equals_decode(nat, real, oracle_model):
    """gecco_crf_plan_run_decode_pipelined: call k carries the marginals of batch k and the Viterbi labels of batch k - 1
    (one launch when both qualify: crf_decode_pipelined).  Same bits as gecco_crf_plan_run_decode for every batch, over a
    sequence that mixes qualifying batches, one plan following itself, a batch with skipped contigs (pad=False), a batch
    with a long contig, an empty batch and a 3-label model (separate launches there)."""
    rng = np.random.default_rng(321)
    A = oracle_model["state"].shape[0]
    w3, t3 = rng.normal(size=(A, 3)), rng.normal(size=(3, 3))
    m3 = nat.Model.from_tables(w3, t3)
    specs = [
        (real, list(rng.integers(1, 400, size=150)), True),
        (real, list(rng.integers(100, 300, size=300)), True),
        (real, list(rng.integers(100, 300, size=300)), True),   # (the same layout: its plan is reused below)
        (real, [5, 30, 7, 200, 19, 21] + list(rng.integers(1, 60, size=100)), False),  # skipped contigs: no hand-over
        (real, list(rng.integers(1, 400, size=100)), True),
        (real, [3000, 50, 70], True),                           # a long contig: the general whole-contig path
        (real, [], True),
        (m3, list(rng.integers(1, 100, size=40)), True),
        (real, list(rng.integers(1, 2048, size=30)), True),
        (real, list(rng.integers(1, 400, size=200)), True),
    ]
    batches = []
    for model, lengths, pad in specs:
        cptr, gptr, attr = synth_contigs(rng, lengths, A)
        batches.append((model, cptr, pad, *_dev(gptr, attr), int(cptr[-1])))
    plans = [nat.Plan(m, cptr, 20, 1, pad, device=0) for m, cptr, pad, _, _, _ in batches]
    plans[2] = plans[1]  # one plan following itself: the score differences are double-buffered
    batches[2] = (batches[1][0], batches[1][1], batches[1][2], batches[2][3], batches[2][4], batches[1][5])
    # batch 2 needs CSR arrays of batch 1's shape: rebuild them for that layout
    c2, g2, a2 = synth_contigs(np.random.default_rng(77), np.diff(batches[1][1]), A)
    batches[2] = (batches[1][0], c2, batches[1][2], *_dev(g2, a2), int(c2[-1]))
    exp = []
    for (model, cptr, pad, d_gp, d_at, n), plan in zip(batches, plans):
        p = torch.zeros(max(n, 1), dtype=torch.float64, device="cuda:0")
        y = torch.full((max(n, 1),), 5, dtype=torch.int8, device="cuda:0")
        nat.Plan(model, cptr, 20, 1, pad, device=0).run_decode(d_gp.data_ptr() if n else 0, d_at.data_ptr() if n else 0, p.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        exp.append((p.cpu().numpy()[:n], y.cpu().numpy()[:n]))
    stream = torch.cuda.Stream(device="cuda:0")
    outs = [(torch.zeros(max(b[5], 1), dtype=torch.float64, device="cuda:0"), torch.full((max(b[5], 1),), 5, dtype=torch.int8, device="cuda:0"))
            for b in batches]
    with torch.cuda.stream(stream):
        for k, ((model, cptr, pad, d_gp, d_at, n), plan) in enumerate(zip(batches, plans)):
            plan.run_decode_pipelined(d_gp.data_ptr() if n else 0, d_at.data_ptr() if n else 0, outs[k][0].data_ptr(),
                                      plans[k - 1] if k else None, outs[k - 1][1].data_ptr() if k else 0, 1, stream.cuda_stream)
        plans[-1].flush_decode_pipelined(outs[-1][1].data_ptr(), stream.cuda_stream)
    stream.synchronize()
    for k, ((p, y), (ep, ey)) in enumerate(zip(outs, exp)):
        n = batches[k][5]
        got_p, got_y = p.cpu().numpy()[:n], y.cpu().numpy()[:n]
        assert np.array_equal(np.isnan(got_p), np.isnan(ep)), k
        ok = ~np.isnan(ep)
        assert np.array_equal(got_p[ok], ep[ok]), k
        assert np.array_equal(got_y, ey), k


def test_pipelined_decode_ties_take_crfsuites_recursion(nat):
    """Integer-valued weights make exact ties common: the Viterbi workgroups of the pipelined launch send the contigs that
    hold one through CRFsuite's own recursion (vd_short's exact pass, same body) -- labels equal the oracle's."""
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(99)
    A = 30
    w = rng.integers(-2, 3, size=(A, 2)).astype(np.float64)
    trans = np.array([[1.0, -1.0], [-1.0, 1.0]])
    model = nat.Model.from_tables(w, trans)
    ys, eys = [], []
    plans, bufs = [], []
    for k in range(3):
        cptr, gptr, attr = synth_contigs(rng, list(rng.integers(1, 300, size=120)), A)
        n = int(cptr[-1])
        d_gp, d_at = _dev(gptr, attr)
        plans.append(nat.Plan(model, cptr, 20, 1, True, device=0))
        bufs.append((d_gp, d_at, torch.zeros(n, dtype=torch.float64, device="cuda:0"), torch.full((n,), 5, dtype=torch.int8, device="cuda:0")))
        eys.append(orc.viterbi(w, trans, cptr, gptr, attr)[0])
    for k in range(3):
        plans[k].run_decode_pipelined(bufs[k][0].data_ptr(), bufs[k][1].data_ptr(), bufs[k][2].data_ptr(),
                                      plans[k - 1] if k else None, bufs[k - 1][3].data_ptr() if k else 0)
    plans[2].flush_decode_pipelined(bufs[2][3].data_ptr())
    torch.cuda.synchronize()
    for k in range(3):
        assert np.array_equal(bufs[k][3].cpu().numpy().astype(np.int32), eys[k]), k


def test_two_pipelined_decode_streams_side_by_side(nat, real, oracle_model):
    """bench.py's default schedule: batches alternate between two independent decode streams (a plan and a HIP stream each),
    so that two launches of crf_decode_pipelined are in flight.  Every batch's outputs equal the two-launch decode's."""
    rng = np.random.default_rng(808)
    A = oracle_model["state"].shape[0]
    batches = [synth_contigs(rng, list(rng.integers(50, 400, size=400)), A) for _ in range(6)]
    devb = [_dev(g, a) for _, g, a in batches]
    exp = []
    for (cptr, _, _), (d_gp, d_at) in zip(batches, devb):
        n = int(cptr[-1])
        p = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        y = torch.zeros(n, dtype=torch.int8, device="cuda:0")
        nat.Plan(real, cptr, 20, 1, True, device=0).run_decode(d_gp.data_ptr(), d_at.data_ptr(), p.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        exp.append((p.cpu().numpy(), y.cpu().numpy()))
    plans = [nat.Plan(real, c, 20, 1, True, device=0) for c, _, _ in batches]
    outs = [(torch.zeros(int(c[-1]), dtype=torch.float64, device="cuda:0"), torch.full((int(c[-1]),), 5, dtype=torch.int8, device="cuda:0"))
            for c, _, _ in batches]
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()
    last = [None, None]  # the batch each stream scored last
    for rep in range(3):  # (the same plans again: a plan following itself two calls later on its stream)
        for k in range(6):
            s = k % 2
            prev = last[s]
            plans[k].run_decode_pipelined(devb[k][0].data_ptr(), devb[k][1].data_ptr(), outs[k][0].data_ptr(),
                                          plans[prev] if prev is not None else None, outs[prev][1].data_ptr() if prev is not None else 0, 1,
                                          streams[s].cuda_stream)
            last[s] = k
    for s in range(2):
        plans[last[s]].flush_decode_pipelined(outs[last[s]][1].data_ptr(), streams[s].cuda_stream)
    torch.cuda.synchronize()
    for k in range(6):
        assert np.array_equal(outs[k][0].cpu().numpy(), exp[k][0]) and np.array_equal(outs[k][1].cpu().numpy(), exp[k][1]), k


def test_decode_stream_helper(nat, real, oracle_model):
    """`_native.DecodeStream`: submit / flush bookkeeping of the pipelined decode; two of them side by side."""
    rng = np.random.default_rng(515)
    A = oracle_model["state"].shape[0]
    batches = [synth_contigs(rng, list(rng.integers(20, 300, size=200)), A) for _ in range(5)]
    devb = [_dev(g, a) for _, g, a in batches]
    plans = [nat.Plan(real, c, 20, 1, True, device=0) for c, _, _ in batches]
    outs = [(torch.zeros(int(c[-1]), dtype=torch.float64, device="cuda:0"), torch.full((int(c[-1]),), 5, dtype=torch.int8, device="cuda:0"))
            for c, _, _ in batches]
    exp = []
    for k, (c, _, _) in enumerate(batches):
        p = torch.zeros(int(c[-1]), dtype=torch.float64, device="cuda:0")
        y = torch.zeros(int(c[-1]), dtype=torch.int8, device="cuda:0")
        nat.Plan(real, c, 20, 1, True, device=0).run_decode(devb[k][0].data_ptr(), devb[k][1].data_ptr(), p.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        exp.append((p.cpu().numpy(), y.cpu().numpy()))
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
    ds = [nat.DecodeStream(s.cuda_stream) for s in streams]
    torch.cuda.synchronize()
    for k in range(5):
        ds[k % 2].submit(plans[k], devb[k][0].data_ptr(), devb[k][1].data_ptr(), outs[k][0].data_ptr(), outs[k][1].data_ptr())
    for d in ds:
        d.flush()
        d.flush()  # (nothing left: a no-op)
    torch.cuda.synchronize()
    for k in range(5):
        assert np.array_equal(outs[k][0].cpu().numpy(), exp[k][0]) and np.array_equal(outs[k][1].cpu().numpy(), exp[k][1]), k
