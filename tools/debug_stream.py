"""Where does the streaming window kernel differ from the tiled one?  (run on the GPU box)"""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
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
    np.save(sys.argv[2], got)
    np.save(sys.argv[2] + ".cptr.npy", cptr)
    sys.exit(0)
outs = {}
for tag, env in (("tiled", "0"), ("stream", "4")):
    e = dict(os.environ, GECCO_CRF_STREAM=env)
    subprocess.check_call([sys.executable, __file__, "child", f"/tmp/dbg_{tag}.npy"], env=e)
    outs[tag] = np.load(f"/tmp/dbg_{tag}.npy")
cptr = np.load("/tmp/dbg_tiled.npy.cptr.npy")
d = np.abs(np.nan_to_num(outs["tiled"]) - np.nan_to_num(outs["stream"]))
bad = np.nonzero(d > 1e-12)[0]
print("genes", len(d), "bad", len(bad), "max", d.max())
if len(bad):
    print("first bad genes", bad[:40])
    c = np.searchsorted(cptr, bad, side="right") - 1
    print("their contigs", c[:40], "lengths", np.diff(cptr)[c[:40]], "pos in contig", (bad - cptr[c])[:40])
    # slot space: contigs shorter than W padded to W
    n = np.diff(cptr); np_ = np.maximum(n, 20); cslot = np.concatenate([[0], np.cumsum(np_)])
    slot = cslot[c] + (np_[c] - n[c]) // 2 + (bad - cptr[c])
    print("slots", slot[:40], "wg (4 phases)", (slot // 1005)[:40], "r-index", ((slot % 1005) + 19)[:40])
    for g in bad[:10]:
        print(g, outs["tiled"][g], outs["stream"][g])
    # per workgroup: bad genes, and whether a padded contig lies in its reach
    OUTW = 1005
    nwg = (cslot[-1] + OUTW - 1) // OUTW
    allslot = np.concatenate([cslot[k] + (np_[k] - n[k]) // 2 + np.arange(n[k]) for k in range(len(n))])
    badmask = d > 1e-12
    for w in range(nwg):
        lo, hi = w * OUTW - 19, w * OUTW + OUTW + 19
        inreach = [(k, n[k]) for k in range(len(n)) if cslot[k + 1] > lo and cslot[k] < hi]
        padded = [k for k, nk in inreach if nk < 20]
        sel = (allslot >= w * OUTW) & (allslot < (w + 1) * OUTW)
        bs = allslot[sel & badmask]
        print("wg", w, "contigs", inreach[0][0], "..", inreach[-1][0], "padded", padded, "bad", int((sel & badmask).sum()),
              "first bad r-index", (int(bs.min()) - w * OUTW + 19) if len(bs) else None, "last", (int(bs.max()) - w * OUTW + 19) if len(bs) else None)
