"""No kernel writes outside the arrays it is given: every output of the resident API sits between two guard bands of
a recognisable pattern (and starts at an odd element, so that 8- and 16-byte stores have no alignment to lean on);
after each launch the bands must be untouched.  Shapes: contigs shorter than the window (padding), a contig longer
than a scan block, sizes that are no multiple of anything."""
import numpy as np
import pytest

import torch

from tests.helpers import synth_contigs

pytestmark = pytest.mark.gpu

GUARD = 96  # bytes either side


class Guarded:
    def __init__(self, n_items, dtype, odd_bytes):
        item = torch.empty(0, dtype=dtype).element_size()
        self.nbytes = max(int(n_items), 1) * item
        self.lead = GUARD + odd_bytes
        self.raw = torch.full((self.lead + self.nbytes + GUARD,), 0xA5, dtype=torch.uint8, device="cuda:0")
        self.dtype, self.n = dtype, int(n_items)

    @property
    def ptr(self):
        return self.raw.data_ptr() + self.lead

    def check(self, what):
        raw = self.raw.cpu().numpy()
        assert (raw[: self.lead] == 0xA5).all(), f"{what}: written before the array"
        assert (raw[self.lead + self.nbytes:] == 0xA5).all(), f"{what}: written past the array"

    def values(self):
        raw = self.raw.cpu().numpy()[self.lead: self.lead + self.nbytes].copy()
        return raw.view(torch.empty(0, dtype=self.dtype).numpy().dtype)[: self.n]


@pytest.mark.parametrize("lengths,pad", [([5, 19, 1, 7, 20, 21], True), ([333, 2, 2500, 64, 1], True), ([333, 2, 2500, 64, 1], False),
                                         ([4097, 2049, 1], True)])
def test_outputs_stay_inside_their_arrays(oracle_model, lengths, pad):
    import os

    from gecco_amd import _native as nat
    from oracle import crf_oracle as orc
    from oracle import lcrf
    from tests.helpers import GOLDEN

    model = nat.Model.from_lcrf(lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))["blob"])
    rng = np.random.default_rng(len(lengths) + sum(lengths))
    cptr, gptr, attr = synth_contigs(rng, lengths, oracle_model["state"].shape[0])
    n, nc = int(cptr[-1]), len(cptr) - 1
    d_gp = torch.from_numpy(gptr).to("cuda:0")
    d_at = torch.from_numpy(attr if len(attr) else np.zeros(1, np.int32)).to("cuda:0")
    ann = (np.diff(gptr) > 0).astype(np.uint8)
    d_ann = torch.from_numpy(ann).to("cuda:0")
    plan = nat.Plan(model, cptr, 20, 1, pad, device=0)
    # 8-byte aligned (the ABI's contract for double arrays) but not 16-byte aligned; labels at an odd byte
    p, y = Guarded(n, torch.float64, 8), Guarded(n, torch.int8, 3)
    marg, ln, sc = Guarded(2 * n, torch.float64, 8), Guarded(nc, torch.float64, 8), Guarded(nc, torch.float64, 8)
    seg, nseg = Guarded(4 * n, torch.int32, 4), Guarded(1, torch.int32, 4)

    def all_ok(what):
        torch.cuda.synchronize()
        for g, name in ((p, "p"), (y, "y"), (marg, "marginals"), (ln, "log Z"), (sc, "scores"), (seg, "rows"), (nseg, "row count")):
            g.check(f"{what}: {name}")

    plan.run_windowed(d_gp.data_ptr(), d_at.data_ptr(), p.ptr, 1)
    all_ok("windowed")
    p_exp = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, pad)
    got = p.values()
    ok = ~np.isnan(p_exp)
    assert np.abs(got[ok] - p_exp[ok]).max() <= 1e-12
    plan.run_decode(d_gp.data_ptr(), d_at.data_ptr(), p.ptr, y.ptr, 1)
    all_ok("decode")
    ey, _ = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)
    assert np.array_equal(y.values().astype(np.int32), ey)
    plan.run_viterbi(d_gp.data_ptr(), d_at.data_ptr(), y.ptr, sc.ptr)
    all_ok("viterbi with scores")
    plan.run_viterbi(d_gp.data_ptr(), d_at.data_ptr(), y.ptr)
    all_ok("viterbi")
    plan.run_marginals_full(d_gp.data_ptr(), d_at.data_ptr(), marg.ptr, ln.ptr)
    all_ok("whole-contig marginals")
    plan.run_marginals_full(d_gp.data_ptr(), d_at.data_ptr(), marg.ptr)
    all_ok("whole-contig marginals without log Z")
    if pad:
        plan.run_segment(p.ptr, d_ann.data_ptr(), seg.ptr, n, nseg.ptr, 0.5, 2, 0, True)
        all_ok("segment")
        k = int(nseg.values()[0])
        exp = orc.segment(got, ann, cptr, 0.5, 2, 0, True)
        assert seg.values()[: 4 * k].reshape(-1, 4).tolist() == exp.tolist()
