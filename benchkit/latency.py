"""Small-batch latency of the hot path: what ONE call costs at 50 / 1 000 / 10 000 / 100 000 genes.

BASELINE.json configs[0] ("C1") is what `gecco run` on one genome does: one contig of ~50 genes through
`ClusterCRF.predict_probabilities` (/root/reference/gecco/crf/__init__.py:244-258, reached by
tests/test_cli/test_run.py:35-70) with the pretrained weights.  At that size nothing is bandwidth- or issue-bound:
the call is a handful of HIP API calls and one launch, so what is reported here is wall time per call, cold
(first call of a fresh model + session: allocations, table uploads) and warm (median of >= 200 calls), at every
level a caller can enter the path:

  resident       plan + device-resident CSR: `gecco_crf_plan_run_decode` + stream synchronisation (latency form),
                 and the pipelined decode back to back (throughput form, time per batch)
  one_shot       `gecco_crf_windowed_marginals` on pinned host buffers (C ABI time, and the same through the
                 Python binding `Session.windowed_marginals`)
  decode         `gecco_crf_session_decode_wire` (marginals + Viterbi labels, compact wire format)
  clusters_wire  `gecco_crf_session_clusters_wire` (marginals + refiner on the device, rows back)
  object_api     `ClusterCRF.predict_probabilities` on `Gene` objects

Synthetic inputs follow SURVEY.md 8d's C1 law on the REAL model (tests/golden/model.pkl): distinct domains per
gene {0: .30, 1: .35, 2: .17, >= 3: .18 as 3 + Geometric(.5)}, ids Zipf(1.2) over the model's 2 659 attributes;
50 genes = one contig, larger sizes = contigs of 200 genes.

`python -m benchkit.latency --cold-process` prints what a FRESH process pays before its first result (library load,
runtime start, model parse, session, first call): bench.py runs it as a child process.
"""
import ctypes
import json
import os
import sys
import time
import warnings

import numpy as np

from gecco_amd import _native as nat
from gecco_amd import pickle_model, synth

SIZES = (50, 1000, 10000, 100000)
W, STEP, LABEL = 20, 1, 1


def golden_dir():
    return os.environ.get("GECCO_AMD_MODEL_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def real_blob(model_dir=None):
    return pickle_model.crfsuite_blob(pickle_model.load_model_dir(model_dir or golden_dir()))


def c1_batch(n_genes, num_attrs, seed=synth.SEED):
    """SURVEY.md 8d C1 law: one contig of `n_genes` genes when n_genes <= 200, contigs of 200 genes beyond."""
    rng = np.random.default_rng(seed + n_genes)
    lengths = [n_genes] if n_genes <= 200 else [200] * (n_genes // 200) + ([n_genes % 200] if n_genes % 200 else [])
    return synth.synth_contigs(rng, lengths, num_attrs)


def dedup_csr(gptr, attr):
    """The batch as the object path sees it: a gene's repeated domain names are ONE feature (dict keys,
    /root/reference/gecco/crf/features.py:31-35), first occurrence kept."""
    g2, a2 = [0], []
    for g in range(len(gptr) - 1):
        seen = []
        for a in attr[gptr[g]:gptr[g + 1]].tolist():
            if a not in seen:
                seen.append(a)
        a2.extend(seen)
        g2.append(len(a2))
    return np.asarray(g2, dtype=np.int32), np.asarray(a2, dtype=np.int32)


def _median_us(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    ts = np.empty(reps)
    for i in range(reps):
        t0 = time.perf_counter()
        fn()
        ts[i] = time.perf_counter() - t0
    ts.sort()
    return {"median_us": float(ts[reps // 2] * 1e6), "p10_us": float(ts[reps // 10] * 1e6), "p90_us": float(ts[(reps * 9) // 10] * 1e6),
            "calls": int(reps)}


def _reps_for(n):
    return 400 if n <= 1000 else 200


class _Pinned:
    """The batch in pinned host buffers + the compact wire format's arrays + output buffers."""

    def __init__(self, cptr, gptr, attr):
        self.n = int(cptr[-1])
        self.nc = len(cptr) - 1
        self.cp, self.gp = nat.pinned_copy(cptr), nat.pinned_copy(gptr)
        self.at = nat.pinned_copy(attr if len(attr) else np.zeros(1, np.int32))
        self.at16 = nat.pinned_copy(attr if len(attr) else np.zeros(1, np.int32), np.uint16)
        self.deg = nat.pinned_copy(nat.degree_bytes(gptr))
        self.p = nat.pinned_empty(max(self.n, 1), np.float64)
        self.y = nat.pinned_empty(max(self.n, 1), np.int8)


def one_shot_levels(model, cptr, gptr, attr, device=0, reps=None):
    """Warm one-shot calls on pinned buffers, C ABI time (ctypes call with prepared arguments) and through the Python binding."""
    lib = nat.load_library()
    b = _Pinned(cptr, gptr, attr)
    reps = reps or _reps_for(b.n)
    out = {}
    a = lambda x: x.ctypes.data  # noqa: E731
    vp = ctypes.c_void_p

    # gecco_crf_windowed_marginals: the one-shot entry point of the ABI (the model's own per-device session)
    fn = lib.gecco_crf_windowed_marginals
    args = (model._h, device, b.cp.ctypes.data_as(nat._c_i32p), b.nc, b.gp.ctypes.data_as(nat._c_i32p), b.at.ctypes.data_as(nat._c_i32p),
            W, STEP, LABEL, 1, b.p.ctypes.data_as(nat._c_f64p))

    def call_abi():
        rc = fn(*args)
        if rc:
            nat._check(rc)

    out["one_shot_pinned"] = {**_median_us(call_abi, reps), "entry": "gecco_crf_windowed_marginals, ctypes call with prepared arguments"}
    p_abi = b.p[: b.n].copy()
    ses = nat.Session(model, [device])
    out["one_shot_pinned_python"] = {**_median_us(lambda: ses.windowed_marginals(b.cp, b.gp, b.at, W, out=b.p), reps),
                                     "entry": "Session.windowed_marginals (the Python binding: argument conversion + the same C call)"}
    # pageable numpy arrays in, a fresh numpy array out: what Model.windowed_marginals / ClusterCRF._score do
    cp0, gp0, at0 = np.array(cptr), np.array(gptr), np.array(attr if len(attr) else np.zeros(1, np.int32))
    out["one_shot_pageable_python"] = {**_median_us(lambda: ses.windowed_marginals(cp0, gp0, at0, W), reps),
                                       "entry": "Session.windowed_marginals on pageable numpy arrays, fresh output array"}
    # marginals + labels on the compact wire format
    fn_d = lib.gecco_crf_session_decode_wire
    args_d = (ses._h, vp(a(b.cp)), b.nc, vp(a(b.gp)), vp(a(b.deg)), None, vp(a(b.at16)), W, STEP, LABEL, 1, vp(a(b.p)), vp(a(b.y)))

    def call_decode():
        rc = fn_d(*args_d)
        if rc:
            nat._check(rc)

    out["decode_wire"] = {**_median_us(call_decode, reps), "entry": "gecco_crf_session_decode_wire (marginals + Viterbi labels), pinned buffers"}
    y_abi = b.y[: b.n].copy()
    # cluster calls: rows only
    cap = min(b.n, b.n // 2 + b.nc) + 1
    seg = np.empty((cap, 4), dtype=np.int32)
    seg_off = np.zeros(cap + 1, dtype=np.int64)
    n_seg = ctypes.c_int32(0)
    q = nat.refine_params("gecco", 0.8, 3, 5, 0.6, 0, True, False)
    fn_c = lib.gecco_crf_session_clusters_wire
    args_c = (ses._h, vp(a(b.cp)), b.nc, vp(a(b.gp)), vp(a(b.deg)), None, vp(a(b.at16)), None, W, STEP, LABEL, 1, ctypes.addressof(q),
              None, vp(a(seg)), cap, ctypes.addressof(n_seg), None, max(b.n, 1), vp(a(seg_off)))

    def call_clusters():
        rc = fn_c(*args_c)
        if rc:
            nat._check(rc)

    out["clusters_wire"] = {**_median_us(call_clusters, reps), "clusters": None,
                            "entry": "gecco_crf_session_clusters_wire (marginals + refiner on the device, rows only), pinned buffers"}
    out["clusters_wire"]["clusters"] = int(n_seg.value)
    for v in out.values():
        v["genes"] = b.n
    return out, p_abi, y_abi, seg[: n_seg.value].copy()


def cold_levels(blob, cptr, gptr, attr, device=0):
    """First call of a FRESH model + session in a process whose HIP runtime is already up: what the first result costs by
    component (model parse, session create, first call = device allocations + weight-table upload + launch), then the
    second call (allocations done) and a warm one."""
    lib = nat.load_library()
    b = _Pinned(cptr, gptr, attr)
    t0 = time.perf_counter()
    model = nat.Model.from_lcrf(blob)
    t1 = time.perf_counter()
    ses = nat.Session(model, [device])
    t2 = time.perf_counter()
    ses.windowed_marginals(b.cp, b.gp, b.at, W, out=b.p)
    t3 = time.perf_counter()
    ses.windowed_marginals(b.cp, b.gp, b.at, W, out=b.p)
    t4 = time.perf_counter()
    for _ in range(5):
        ses.windowed_marginals(b.cp, b.gp, b.at, W, out=b.p)
    t5 = time.perf_counter()
    ses.windowed_marginals(b.cp, b.gp, b.at, W, out=b.p)
    t6 = time.perf_counter()
    del lib
    return {"genes": b.n, "model_parse_us": (t1 - t0) * 1e6, "session_create_us": (t2 - t1) * 1e6, "first_call_us": (t3 - t2) * 1e6,
            "second_call_us": (t4 - t3) * 1e6, "warm_call_us": (t6 - t5) * 1e6,
            "note": "fresh model + session, HIP runtime and code objects already loaded by this process: first_call = device "
                    "allocations (lane buffers, plan arena) + weight tables upload + launch + download; Session.windowed_marginals, pinned buffers"}


def resident_levels(model, cptr, gptr, attr, device=0, reps=None):
    """Device-resident CSR: the decode step as ONE synchronous call (two launches + stream synchronisation) and the
    pipelined decode back to back (time per batch)."""
    import torch

    dev = torch.device("cuda", device)
    n = int(cptr[-1])
    reps = reps or _reps_for(n)
    d_gp = torch.from_numpy(np.ascontiguousarray(gptr)).to(dev)
    d_at = torch.from_numpy(np.ascontiguousarray(attr) if len(attr) else np.zeros(1, np.int32)).to(dev)
    d_p = torch.zeros(max(n, 1), dtype=torch.float64, device=dev)
    d_y = torch.zeros(max(n, 1), dtype=torch.int8, device=dev)
    plan = nat.Plan(model, cptr, W, STEP, True, device=device)
    st = torch.cuda.Stream(dev)
    s = st.cuda_stream
    a_gp, a_at, a_p, a_y = d_gp.data_ptr(), d_at.data_ptr(), d_p.data_ptr(), d_y.data_ptr()

    def sync_decode():
        plan.run_decode(a_gp, a_at, a_p, a_y, LABEL, 0, s)
        st.synchronize()

    def sync_windowed():
        plan.run_windowed(a_gp, a_at, a_p, LABEL, s)
        st.synchronize()

    out = {"resident_decode_sync": {**_median_us(sync_decode, reps), "entry": "gecco_crf_plan_run_decode + stream synchronisation"},
           "resident_windowed_sync": {**_median_us(sync_windowed, reps), "entry": "gecco_crf_plan_run_windowed + stream synchronisation"}}
    # throughput form: pipelined decode, the plan following itself, K calls + flush between synchronisations
    K = 200
    plan.run_decode_pipelined(a_gp, a_at, a_p, None, 0, LABEL, s)
    plan.flush_decode_pipelined(a_y, s)
    st.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        plan.run_decode_pipelined(a_gp, a_at, a_p, None, 0, LABEL, s)
        for _ in range(K - 1):
            plan.run_decode_pipelined(a_gp, a_at, a_p, plan, a_y, LABEL, s)
        plan.flush_decode_pipelined(a_y, s)
        st.synchronize()
        ts.append((time.perf_counter() - t0) / K)
    out["resident_step_pipelined"] = {"median_us": float(sorted(ts)[len(ts) // 2] * 1e6), "calls": K,
                                      "entry": "gecco_crf_plan_run_decode_pipelined back to back on one stream, time per batch"}
    for v in out.values():
        v["genes"] = n
    return out, d_p[:n].cpu().numpy(), d_y[:n].cpu().numpy()


def object_level(n_genes, model_dir=None, reps=None, seed=synth.SEED):
    """`ClusterCRF.predict_probabilities` on Gene objects built from the same C1 law."""
    from gecco_amd.crf import ClusterCRF
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand

    crf = ClusterCRF.trained(model_dir or golden_dir())
    attrs = crf.model.attributes_
    cptr, gptr, attr = c1_batch(n_genes, len(attrs), seed)
    genes = []
    for c in range(len(cptr) - 1):
        src = Source(f"contig_{c:05d}")
        for i, g in enumerate(range(cptr[c], cptr[c + 1])):
            doms = [Domain(attrs[a], 10 * j + 1, 10 * j + 9, "Pfam", 1e-10, 1e-12) for j, a in enumerate(attr[gptr[g]:gptr[g + 1]])]
            genes.append(Gene(src, 1000 * i, 1000 * i + 900, Strand.Coding, Protein(f"c{c:05d}_{i}", None, doms)))
    reps = reps or (200 if n_genes <= 1000 else 20 if n_genes <= 10000 else 5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        first = crf.predict_probabilities(genes)
        cold = time.perf_counter() - t0
        r = _median_us(lambda: crf.predict_probabilities(genes), reps, warm=2)
    p = np.array([g._probability for g in first], dtype=np.float64)
    g2, a2 = dedup_csr(gptr, attr)
    same = bool(np.array_equal(p, crf._session().windowed_marginals(cptr, g2, a2 if len(a2) else np.zeros(1, np.int32), W)))
    return {**r, "genes": len(genes), "first_call_us": cold * 1e6, "equals_csr_call_on_deduplicated_batch": same,
            "entry": "ClusterCRF.predict_probabilities on Gene objects: sort + pack + one-shot ABI + new Gene / Domain objects "
                     "(first_call_us: the object's first call, session creation and device allocations included)"}, p, (cptr, gptr, attr)


def latency_block(device=0, sizes=SIZES, with_objects=True, with_resident=True, model_dir=None):
    """The `latency` object of the bench line.  Returns (block, checks): `checks[n]` = inputs and outputs of every size, for the
    caller's parity check against the CPU oracle (bench.py's cpu_baseline leg; nothing here imports the oracle)."""
    blob = real_blob(model_dir)
    model = nat.Model.from_lcrf(blob)
    A = model.num_attrs
    block, checks = {}, {}
    for n in sizes:
        cptr, gptr, attr = c1_batch(n, A)
        entry = {"genes": int(cptr[-1]), "contigs": len(cptr) - 1, "domain_hits": int(gptr[-1])}
        lv, p_abi, y_abi, seg = one_shot_levels(model, cptr, gptr, attr, device)
        entry.update(lv)
        if with_resident:
            rv, p_res, y_res = resident_levels(model, cptr, gptr, attr, device)
            entry.update(rv)
            entry["resident_equals_one_shot"] = bool(np.array_equal(p_res, p_abi) and np.array_equal(y_res, y_abi))
        if with_objects:
            ov, p_obj, _ = object_level(n, model_dir)
            entry["object_api"] = ov
        entry["cold"] = cold_levels(blob, cptr, gptr, attr, device)
        block[str(n)] = entry
        checks[n] = {"cptr": cptr, "gptr": gptr, "attr": attr, "p": p_abi, "y": y_abi, "seg": seg}
    return block, checks, model


def cold_process(device=0, n_genes=50):
    """Everything a fresh process pays before its first result (run as `python -m benchkit.latency --cold-process`)."""
    t = [time.perf_counter()]
    lib = nat.load_library()
    t.append(time.perf_counter())
    nd = nat.device_count()  # (starts the HIP runtime)
    t.append(time.perf_counter())
    blob = real_blob()
    t.append(time.perf_counter())
    model = nat.Model.from_lcrf(blob)
    t.append(time.perf_counter())
    cptr, gptr, attr = c1_batch(n_genes, model.num_attrs)
    ses = nat.Session(model, [device])
    t.append(time.perf_counter())
    p1 = ses.windowed_marginals(cptr, gptr, attr, W)
    t.append(time.perf_counter())
    p2 = ses.windowed_marginals(cptr, gptr, attr, W)
    t.append(time.perf_counter())
    del lib
    us = [(b - a) * 1e6 for a, b in zip(t[:-1], t[1:])]
    return {"genes": int(cptr[-1]), "devices": nd, "dlopen_us": us[0], "hip_runtime_start_us": us[1], "unpickle_md5_us": us[2],
            "model_parse_us": us[3], "session_create_us": us[4], "first_call_us": us[5], "second_call_us": us[6],
            "same_result": bool(np.array_equal(p1, p2)),
            "hip_runtime": next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln), None),
            "note": "a fresh Python process without torch: first_call = code object load + device allocations + weight tables + "
                    "launch; pageable numpy buffers through Session.windowed_marginals.  Where a PyTorch wheel is installed the "
                    "library loads the wheel's copy of the HIP runtime (so that a later `import torch` still finds the device): "
                    "larger libraries, ~70 ms more; GECCO_AMD_HIP_RUNTIME=system keeps the system's copy"}


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--cold-process", action="store_true")
    ap.add_argument("--sizes", default=",".join(str(s) for s in SIZES))
    ap.add_argument("--no-objects", action="store_true")
    ap.add_argument("--no-resident", action="store_true")
    ap.add_argument("--loop", type=int, default=0, help="N warm one-shot C1 calls and nothing else (for rocprofv3 timelines)")
    ap.add_argument("--loop-entry", default="windowed", choices=["windowed", "decode", "clusters"])
    ap.add_argument("--loop-genes", type=int, default=50)
    ap.add_argument("--loop-sleep-ms", type=float, default=2.0, help="idle time between the calls of --loop (separates them in a timeline)")
    args = ap.parse_args(argv)
    if args.cold_process:
        print(json.dumps(cold_process()))
        return
    if args.loop:
        model = nat.Model.from_lcrf(real_blob())
        cptr, gptr, attr = c1_batch(args.loop_genes, model.num_attrs)
        b = _Pinned(cptr, gptr, attr)
        ses = nat.Session(model, [0])
        for _ in range(args.loop):
            time.sleep(args.loop_sleep_ms * 1e-3)
            if args.loop_entry == "windowed":
                ses.windowed_marginals(b.cp, b.gp, b.at, W, out=b.p)
            elif args.loop_entry == "decode":
                ses.decode(b.cp, b.gp, b.at16, W, out_p=b.p, out_y=b.y, degree=b.deg)
            else:
                ses.clusters(b.cp, b.gp, b.at16, None, W, want_p=False, want_seg_p=False, degree=b.deg)
        return
    sizes = tuple(int(s) for s in args.sizes.split(",") if s)
    if not args.no_resident:
        import torch  # noqa: F401  (before libgecco_crf.so: the wheel's own HIP runtime has to be the first one loaded)
    block, _, _ = latency_block(sizes=sizes, with_objects=not args.no_objects, with_resident=not args.no_resident)
    print(json.dumps(block))


if __name__ == "__main__":
    main(sys.argv[1:])
