"""Diagnostic: cluster-call level before / after a decode call on the same session; host-side trace of one call."""
import os, sys, time
import numpy as np
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gecco_amd import _native as nat, synth

wl = synth.workload("C3", law="8d")
model = nat.Model.from_tables(wl["w"], wl["trans"])
n = int(wl["contig_ptr"][-1])
cp, gp, at = nat.pinned_copy(wl["contig_ptr"]), nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
ann = nat.pinned_copy((np.diff(wl["gene_ptr"]) > 0).astype(np.uint8))
outp, outy = nat.pinned_empty(n, np.float64), nat.pinned_empty(n, np.int8)
deg = nat.pinned_copy(nat.degree_bytes(wl["gene_ptr"]))

def t(fn, reps=5):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3

ses = nat.Session(model, [0])
print("clusters (fresh session)   %.3f ms" % t(lambda: ses.clusters(cp, gp, at, ann, 20, want_p=False, want_seg_p=True)))
print("windowed pinned            %.3f ms" % t(lambda: ses.windowed_marginals(cp, gp, at, 20, out=outp)))
print("windowed degree bytes      %.3f ms" % t(lambda: ses.windowed_marginals(cp, gp, at, 20, out=outp, degree=deg)))
print("clusters again             %.3f ms" % t(lambda: ses.clusters(cp, gp, at, ann, 20, want_p=False, want_seg_p=True)))
print("decode pinned              %.3f ms" % t(lambda: ses.decode(cp, gp, at, 20, out_p=outp, out_y=outy)))
print("clusters after decode      %.3f ms" % t(lambda: ses.clusters(cp, gp, at, ann, 20, want_p=False, want_seg_p=True)))
print("clusters, no seg_p         %.3f ms" % t(lambda: ses.clusters(cp, gp, at, ann, 20, want_p=False, want_seg_p=False)))
seg, seg_p, seg_off, _ = ses.clusters(cp, gp, at, ann, 20, want_p=False, want_seg_p=True)
print("clusters", len(seg), "genes in clusters", int(seg_off[-1]), "stats", ses.stats())
import ctypes
os.environ["GECCO_CRF_TRACE"] = "1"
t0 = time.perf_counter(); ses.clusters(cp, gp, at, ann, 20, want_p=False, want_seg_p=True); print("one call %.3f ms" % ((time.perf_counter() - t0) * 1e3))
