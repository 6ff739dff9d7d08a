"""Debug builds of the streaming kernel: slot constants (dbg1) / start flags (dbg2) against host values."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import synth_contigs, GOLDEN
from gecco_amd import _native as nat
from oracle import lcrf
st = lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))
om = lcrf.load_model(os.path.join(GOLDEN, "model.pkl"), os.path.join(GOLDEN, "model.pkl.md5"))
model = nat.Model.from_lcrf(st["blob"])
W, step, label = 20, 1, 1
rng = np.random.default_rng(1000 * W + 10 * step + label)
lengths = [1, 2, 3, W - 1, W, W + 1, 2 * W, 19, 20, 21, 39, 40, 41, 236, 237, 238, 255, 256, 257,
           274, 275, 276, 474, 475, 513, 1000] + list(rng.integers(1, 400, size=60))
rng.shuffle(lengths)
cptr, gptr, attr = synth_contigs(rng, lengths, om["state"].shape[0])
got = model.windowed_marginals(cptr, gptr, attr, W, step, label, True)
w, tr = om["state"], om["trans"]
dw = w[:, 1] - w[:, 0]
cs = np.concatenate([[0.0], np.cumsum(dw[attr])])
d = cs[gptr[1:]] - cs[gptr[:-1]]
mu01 = np.exp(tr[0, 1] + tr[1, 0] - 2 * tr[0, 0])
r = mu01 * np.exp(d)
n = np.diff(cptr); np_ = np.maximum(n, 20); cslot = np.concatenate([[0], np.cumsum(np_)])
allslot = np.concatenate([cslot[k] + (np_[k] - n[k]) // 2 + np.arange(n[k]) for k in range(len(n))])
start = np.zeros(len(d)); 
for k in range(len(n)):
    if n[k] >= 20: start[cptr[k]:cptr[k] + n[k] - 19] = 1
mode = sys.argv[1]
exp = r if mode == "1" else start
rel = np.abs(got - exp) / np.maximum(np.abs(exp), 1e-300) if mode == "1" else np.abs(got - exp)
# genes of padded contigs have starts in padding slots: ignore them for mode 2
bad = np.nonzero((rel > 1e-9) & (n[np.searchsorted(cptr, np.arange(len(d)), side="right") - 1] >= 20))[0]
print("mode", mode, "genes", len(d), "bad", len(bad))
OUTW = 1005
for g in bad[:30]:
    print("gene", g, "slot", allslot[g], "wg", allslot[g] // OUTW, "r-index", allslot[g] % OUTW + 19, "got", got[g], "exp", exp[g])
