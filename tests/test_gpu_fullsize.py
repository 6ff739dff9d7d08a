"""BASELINE.json configurations at their full size under `pytest -m gpu`:
C3 = 10 000 contigs / 2 M genes (the headline workload) and C5 = 100 contigs x 50 000 genes (the
scan-length-bound shape), every output of the path against the CPU oracle: windowed marginals
(<= 1e-12, identical 0.8-calls), Viterbi labels (exact), whole-contig marginals (<= 1e-12), cluster
rows (exact).  About 10 s of oracle time in total."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")  # (before libgecco_crf.so: the wheel's own libamdhip64 has to be the first one loaded)

TOL = 1e-12


def _check_workload(name, whole_contig_marginals):
    from gecco_amd import _native as nat, synth
    from oracle import crf_oracle as orc

    wl = synth.workload(name)
    cptr, gptr, attr, w, trans = wl["contig_ptr"], wl["gene_ptr"], wl["attr_id"], wl["w"], wl["trans"]
    n = int(cptr[-1])
    model = nat.Model.from_tables(w, trans)
    ses = nat.Session(model, [0])
    p, y = ses.decode(cptr, gptr, attr, 20, 1, 1, True)
    exp = orc.windowed_marginals_mt(w, trans, cptr, gptr, attr, 20, 1, 1, True, threads=32)
    assert np.abs(p - exp).max() <= TOL
    assert int(((p > 0.8) != (exp > 0.8)).sum()) == 0
    ey, _ = orc.viterbi(w, trans, cptr, gptr, attr)
    assert int((y.astype(np.int32) != ey).sum()) == 0
    # cluster calls, resident epilogue vs the oracle's refiner on the oracle's probabilities
    ann = (np.diff(gptr) > 0).astype(np.uint8)
    seg, seg_p, seg_off, _ = ses.clusters(cptr, gptr, attr, ann, 20, 1, 1, True, 0.8, 3, 0, True)
    eseg = orc.segment(exp, ann, cptr, 0.8, 3, 0, True, carry_state=False)
    assert len(eseg) > 0 and seg.tolist() == eseg.tolist()
    assert seg_off[-1] == int((seg[:, 3] - seg[:, 2]).sum())
    # the throughput form of the decode step on resident buffers: the plan following itself three times, then the flush
    # (C3: one launch per batch, crf_decode_pipelined; C5: the general whole-contig kernels behind the same call)
    dev = torch.device("cuda:0")
    d_gp, d_at = torch.from_numpy(gptr).to(dev), torch.from_numpy(attr).to(dev)
    plan = nat.Plan(model, cptr, 20, 1, True, device=0)
    ps = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(4)]
    ys = [torch.full((n,), 9, dtype=torch.int8, device=dev) for _ in range(4)]
    plan.run_decode(d_gp.data_ptr(), d_at.data_ptr(), ps[3].data_ptr(), ys[3].data_ptr())  # two launches, the same tiling
    torch.cuda.synchronize()
    p2, y2 = ps[3].cpu().numpy(), ys[3].cpu().numpy()
    assert np.abs(p2 - exp).max() <= TOL and np.array_equal(y2, y)
    for k in range(3):
        plan.run_decode_pipelined(d_gp.data_ptr(), d_at.data_ptr(), ps[k].data_ptr(), plan if k else None, ys[k - 1].data_ptr() if k else 0)
    plan.flush_decode_pipelined(ys[2].data_ptr())
    torch.cuda.synchronize()
    for k in range(3):
        assert np.array_equal(ps[k].cpu().numpy(), p2), k       # the bits of the two-launch decode
        assert np.array_equal(ys[k].cpu().numpy(), y2), k
    del ps, ys, d_gp, d_at, plan
    if whole_contig_marginals:
        marg, ln = model.marginals_full(cptr, gptr, attr)
        em, eln = orc.full_marginals(w, trans, cptr, gptr, attr)
        assert np.abs(marg - em).max() <= TOL
        assert np.abs(ln - eln).max() <= 1e-9 * max(1.0, float(np.abs(eln).max()))
    return n


def test_c3_full_size():
    assert _check_workload("C3", True) == 1_999_989


def test_c5_full_shape():
    assert _check_workload("C5", True) == 5_000_000
