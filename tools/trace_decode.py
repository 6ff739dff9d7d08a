"""One decode call (windowed marginals + Viterbi labels, pinned buffers) per millisecond-separated burst: GECCO_CRF_TRACE=1
prints the batch driver's laps; under `rocprofv3 --kernel-trace --memory-copy-trace` read the result with tools/timeline.py."""
import os, sys, time
import numpy as np
if os.environ.get("NO_TORCH") != "1":  # NO_TORCH=1: the system HIP runtime instead of the one PyTorch bundles
    import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gecco_amd import _native as nat, synth

wl = synth.workload("C3")
model = nat.Model.from_tables(wl["w"], wl["trans"])
n = int(wl["contig_ptr"][-1])
cp, gp, at = nat.pinned_copy(wl["contig_ptr"]), nat.pinned_copy(wl["gene_ptr"]), nat.pinned_copy(wl["attr_id"])
outp, outy = nat.pinned_empty(n, np.float64), nat.pinned_empty(n, np.int8)
ses = nat.Session(model, [0])
if len(sys.argv) > 1:
    ses.set_chunk_genes(int(sys.argv[1]))
wire = os.environ.get("WIRE", "0") == "1"  # WIRE=1: degree bytes + 16-bit attribute indices on the wire
deg = nat.pinned_copy(nat.degree_bytes(wl["gene_ptr"])) if wire else None
if wire:
    at = nat.pinned_copy(wl["attr_id"], np.uint16)
for i in range(6):
    sys.stderr.write("---- call %d\n" % i)
    t0 = time.perf_counter()
    ses.decode(cp, gp, at, 20, out_p=outp, out_y=outy, degree=deg)
    sys.stderr.write("call %d: %.3f ms %s\n" % (i, (time.perf_counter() - t0) * 1e3, ses.stats()))
    time.sleep(0.01)
