#!/usr/bin/env python3
"""Random shapes through the whole-contig kernels against the oracle: contig lengths chosen so that workgroups of
whole contigs end exactly at, just before and just after 2048 genes, partial last lanes, single-gene contigs, one long
contig among short ones (general path).  usage: stress_sequence.py [n_batches] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gecco_amd import _native as nat  # noqa: E402
from oracle import crf_oracle as orc, lcrf  # noqa: E402
from tests.helpers import synth_contigs  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
om = lcrf.load_model(os.path.join(ROOT, "tests", "golden", "model.pkl"), os.path.join(ROOT, "tests", "golden", "model.pkl.md5"))
model = nat.Model.from_lcrf(lcrf.load_pickle(os.path.join(ROOT, "tests", "golden", "model.pkl"))["blob"])
A = om["state"].shape[0]
worst = 0.0
for b in range(n_batches):
    kind = b % 5
    if kind == 0:
        lengths = list(rng.integers(1, 40, size=int(rng.integers(1, 300))))
    elif kind == 1:  # sums landing around the block size
        lengths = [int(x) for x in rng.choice([2047, 2048, 1024, 1023, 1, 2, 7, 8, 9, 1000, 1048], size=int(rng.integers(1, 12)))]
    elif kind == 2:
        lengths = list(rng.integers(100, 2049, size=int(rng.integers(1, 8)))) + [1] * int(rng.integers(0, 20))
    elif kind == 3:  # one long contig: the general path
        lengths = list(rng.integers(1, 300, size=10)) + [int(rng.integers(2049, 9000))] + list(rng.integers(1, 300, size=5))
    else:
        lengths = [int(rng.integers(1, 2049))]
    rng.shuffle(lengths)
    cptr, gptr, attr = synth_contigs(rng, lengths, A)
    y, _ = model.viterbi(cptr, gptr, attr, want_score=False)
    ey, _ = orc.viterbi(om["state"], om["trans"], cptr, gptr, attr)
    assert np.array_equal(y.astype(np.int32), ey), (b, lengths)
    m, ln = model.marginals_full(cptr, gptr, attr)
    em, eln = orc.full_marginals(om["state"], om["trans"], cptr, gptr, attr)
    d = max(np.abs(m - em).max(), np.abs((ln - eln) / np.maximum(1.0, np.abs(eln))).max())
    assert d <= 1e-12, (b, d, lengths)
    worst = max(worst, d)
    p, y2 = nat.Session(model, [0]).decode(cptr, gptr, attr, 20) if b % 10 == 0 else (None, y)
    assert np.array_equal(np.asarray(y2).astype(np.int32), ey)
print("ok", n_batches, "batches; worst |d marginal|, rel |d log Z| =", worst)
