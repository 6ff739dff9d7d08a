"""One cluster call (rows only, degree bytes on the wire) per millisecond-separated burst: run under
`rocprofv3 --kernel-trace --memory-copy-trace` and read with tools/timeline.py."""
import os, sys, time
import numpy as np
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gecco_amd import _native as nat, synth

wl = synth.workload("C3")
model = nat.Model.from_tables(wl["w"], wl["trans"])
n = int(wl["contig_ptr"][-1])
cp, gp, at = nat.pinned_copy(wl["contig_ptr"]), nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
ann = nat.pinned_copy((np.diff(wl["gene_ptr"]) > 0).astype(np.uint8))
deg = nat.pinned_copy(nat.degree_bytes(wl["gene_ptr"]))
ses = nat.Session(model, [0])
if len(sys.argv) > 1:
    ses.set_chunk_genes(int(sys.argv[1]))
for i in range(6):
    t0 = time.perf_counter()
    ses.clusters(cp, gp, at, None, 20, want_p=False, want_seg_p=False, degree=deg)
    print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), ses.stats())
    time.sleep(0.01)
at16 = nat.pinned_copy(wl["attr_id"], np.uint16)
for i in range(6):
    t0 = time.perf_counter()
    ses.clusters(cp, gp, at16, None, 20, want_p=False, want_seg_p=False, degree=deg)
    print("16-bit indices, call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), ses.stats())
    time.sleep(0.01)
for i in range(4):
    t0 = time.perf_counter()
    ses.clusters(cp, gp, at, None, 20, want_p=False, want_seg_p=True, degree=deg)
    print("with probabilities, call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), ses.stats())
