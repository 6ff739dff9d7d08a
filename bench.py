#!/usr/bin/env python3
"""Benchmark of the GECCO CRF hot path on MI355X (contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one resident batch: the windowed forward-backward
marginals of every gene (the reference's `ClusterCRF.predict_probabilities` arithmetic,
gecco/crf/__init__.py:244-258) followed by whole-contig Viterbi decoding of the same batch
.  Workload at N=1 = BASELINE.json configs[2] ("C3": 10k-contig synthetic metagenome, ~2M genes,
35k-attribute synthetic model, W=20).

Schedule of the steps (`--schedule`): "pipelined" (default) is the throughput form of the decode API,
`gecco_crf_plan_run_decode_pipelined`: call k enqueues ONE launch that carries the window tiles of batch k
and the Viterbi workgroups of batch k - 1; the K steps of the timed region are K such calls -- the first
one tiles only -- plus the flush that delivers the labels of the last batch, so every batch's marginals
AND labels are computed inside the region (K window passes, K Viterbi passes, K + 1 launches).
"two-launch" is `gecco_crf_plan_run_decode` (window kernel, then Viterbi kernel, per batch); its step time
is reported next to the headline as `two_launch_ms_per_step`.

N>1 (one process per GPU, no collective on the data path):
  * `value` is WEAK scaling -- every rank owns its own C3-sized batch of contigs;
  * `strong_scaling` in the same JSON line is BASELINE.json configs[3] ("C4"): the ONE C3 batch
    greedy-partitioned by gene count over the N devices (`gecco_amd.sharding.partition_contigs`),
    every rank scoring its own shard.
Inputs are resident in HBM before the timed region.  One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
SIMDS = 256 * 4         # 256 CUs x 4 SIMDs
SCLK_HZ = 2.4e9         # peak engine clock; an fp64 / DPP / 32-bit VALU instruction occupies a SIMD for 4 cycles per wave
VALU_SUSTAINED_CYCLES = 4.7  # ... nominally; measured on MI355X under sustained fp64 load: 4.6-4.8 (tools/ubench/valu_rates.hip)
W, STEP, LABEL = 20, 1, 1


def _alg_bytes(n_genes, nnz, n_contigs):
    # algorithmic bytes of one windowed launch: CSR row pointers + attribute ids in,
    # one fp64 probability per gene out, + the contig table; weight table excluded.
    return 4 * (n_genes + 1) + 4 * nnz + 8 * n_genes + 4 * (n_contigs + 1)


def _ratio_form_fallback_fraction(wl, plan_tiles, tile_out):
    """Fraction of the window kernel's WINDOWS that leave the 3+3-op ratio form of the DP: a wave repeats a phase in the
    max-normalised form when one of its windows ends on Z >= 1e250 (crf_kernels.hip), i.e. roughly when the positive
    score differences of a window add up to more than 575.  Estimated on the host from the weights, for the report only:
    the fraction of waves (64 consecutive window starts) that hold such a window."""
    d_attr = wl["w"][:, LABEL] - wl["w"][:, 1 - LABEL]
    gp = wl["gene_ptr"].astype(np.int64)
    csum = np.concatenate([[0.0], np.cumsum(d_attr[wl["attr_id"]])])
    pos = np.maximum(csum[gp[1:]] - csum[gp[:-1]], 0.0)
    c = np.concatenate([[0.0], np.cumsum(pos)])
    if len(c) <= W:
        return 0.0
    ws = c[W:] - c[:-W]  # window sums (contig boundaries ignored: an over-estimate)
    hot = ws > 575.0
    if not hot.any():
        return 0.0
    return float(np.maximum.reduceat(hot.astype(np.int8), np.arange(0, len(hot), 64)).mean())


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="C3", choices=["C1", "C2", "C3", "C5", "Cinf"])
    ap.add_argument("--no-latency", action="store_true", help="skip the small-batch latency block (50 ... 100 000 genes, cold and warm)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-past-l3", action="store_true", help="skip the HBM-resident (2e8-gene) roofline point")
    ap.add_argument("--kernel-iters", type=int, default=200)
    ap.add_argument("--windowed-only", action="store_true",
                    help="steps are plain windowed launches (for PMC passes: one kind of dispatch)")
    ap.add_argument("--preroll-ms", type=float, default=30.0,
                    help="untimed device pre-roll before the warmup steps: the GPU needs ~10 ms of sustained work to reach steady clocks")
    ap.add_argument("--synth", default="8d", choices=["genome", "8d"],
                    help="weight law of the synthetic model: SURVEY.md 8d to the letter (default) or 'genome' (weights shifted so that "
                         "most genes lean to label 0, as under GECCO's embedded model: the side point `roofline_genome`)")
    ap.add_argument("--min-region-ms", type=float, default=50.0,
                    help="the K-step timed region is repeated until this much time has been measured; ms_per_step is the median region")
    ap.add_argument("--no-levels", action="store_true", help="skip the host-buffer / tables / object API levels (SURVEY.md 8d)")
    ap.add_argument("--no-8d", "--no-genome", dest="no_8d", action="store_true", help="skip the second roofline point on the other weight law")
    ap.add_argument("--no-c4", action="store_true", help="skip the 8-way shard point (profiles: keeps the per-kernel averages on one workload)")
    ap.add_argument("--streams", type=int, default=2,
                    help="pipelined schedule: independent decode streams (plan + HIP stream each) the batches alternate between (three help an "
                         "8-way shard of C3 -- 4.2-4.7 against 5.3-5.9 us -- and nothing else: profiles/EXPERIMENTS.md)")
    ap.add_argument("--schedule", default="pipelined", choices=["pipelined", "two-launch"],
                    help="pipelined: one launch per step = window tiles of batch k + Viterbi workgroups of batch k - 1 "
                         "(gecco_crf_plan_run_decode_pipelined, + one flush); two-launch: gecco_crf_plan_run_decode")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # GECCO_BENCH_ONE_DEVICE=1 (dry runs of the N > 1 code path on a one-GPU box): every rank on device 0, gloo
    one_device = os.environ.get("GECCO_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    red_dev = dev
    backend = None
    # GECCO_BENCH_FORCE_DIST=1: the process group, its probe all-reduce and the gathers also at world size 1 (under
    # `torch.distributed.run --nproc-per-node 1`): what a one-GPU box can exercise of the N > 1 code path, RCCL included
    force_dist = os.environ.get("GECCO_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist

        # RCCL ("nccl") over xGMI when it comes up; gloo otherwise (the collectives here are a barrier and two scalar
        # reductions around the timed region: the data path has none)
        backend = "gloo" if one_device else os.environ.get("GECCO_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            try:
                dist.init_process_group(backend="nccl", device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize(dev)
            except Exception as err:  # every rank fails the same way (no RCCL transport): fall back together
                print(f"[bench rank {rank}] nccl unavailable ({type(err).__name__}: {err}); falling back to gloo", file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                backend = "gloo"
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
        if backend == "gloo":
            dist.init_process_group(backend="gloo")
            red_dev = torch.device("cpu")

    from benchkit import line as bline
    from gecco_amd import _native as nat
    from gecco_amd import sharding, synth

    def barrier():
        if dist is not None:
            dist.barrier()

    lane_streams = {}

    class Resident:
        """One batch resident on this rank's device: plan + CSR + outputs."""

        def __init__(self, model, cptr, gptr, attr, lanes=1):
            self.n_genes, self.nnz, self.n_contigs = int(cptr[-1]), int(gptr[-1]), len(cptr) - 1
            self.d_gp = torch.from_numpy(np.ascontiguousarray(gptr)).to(dev)
            self.d_at = torch.from_numpy(np.ascontiguousarray(attr) if len(attr) else np.zeros(1, np.int32)).to(dev)
            # `lanes` independent decode streams (pipelined schedule): a plan (workspace), a HIP stream and output buffers
            # each; batches alternate between them, so that the tail of one launch overlaps the head of the next
            self.lanes = []
            for k in range(max(1, lanes)):
                ln = {"plan": nat.Plan(model, cptr, W, STEP, True, device=local_rank),
                      "d_p": torch.zeros(max(self.n_genes, 1), dtype=torch.float64, device=dev),
                      "d_y": torch.zeros(max(self.n_genes, 1), dtype=torch.int8, device=dev), "primed": False}
                # a stream of its own, not the legacy default one (streams are captured into graphs everywhere but there)
                if os.environ.get("GECCO_BENCH_OWN_STREAM", "1") == "1" or k > 0:
                    # ONE stream per lane index for the whole process: the batches of this script are stepped one after the other,
                    # and the runtime binds every new stream to one of a few hardware queues -- two streams of a later batch that
                    # land on the same queue serialise (seen as a 9 us shard step next to a 5 us one, by number of streams made before)
                    mask = os.environ.get("GECCO_BENCH_CU_MASK")  # experiment: decode stream k sees a subset of the CUs (profiles/EXPERIMENTS.md)
                    if mask:
                        if k not in lane_streams:
                            lane_streams[k] = _masked_stream(mask, k, max(1, lanes))
                        ln["stream"] = lane_streams[k]
                    else:
                        if k not in lane_streams:
                            lane_streams[k] = torch.cuda.Stream(dev)
                        ln["torch_stream"] = lane_streams[k]
                        ln["stream"] = ln["torch_stream"].cuda_stream
                else:
                    ln["stream"] = torch.cuda.current_stream(dev).cuda_stream
                self.lanes.append(ln)
            l0 = self.lanes[0]
            self.plan, self.d_p, self.d_y, self.stream = l0["plan"], l0["d_p"], l0["d_y"], l0["stream"]
            self.turn = 0
            # device addresses as plain ints: the timed loop is a ctypes call per step and nothing else
            self.a_gp, self.a_at = self.d_gp.data_ptr(), self.d_at.data_ptr()
            for ln in self.lanes:
                ln["a_p"], ln["a_y"] = ln["d_p"].data_ptr(), ln["d_y"].data_ptr()
                ln["call"] = ln["plan"].bind_decode_pipelined(self.a_gp, self.a_at, ln["a_p"], ln["plan"], ln["a_y"], LABEL, ln["stream"])

        def step(self, schedule=None):
            schedule = schedule or args.schedule
            if args.windowed_only:
                self.plan.run_windowed(self.d_gp.data_ptr(), self.d_at.data_ptr(), self.d_p.data_ptr(), LABEL, self.stream)
            elif schedule == "pipelined":  # window tiles of this batch + Viterbi workgroups of the lane's batch before, one launch
                ln = self.lanes[self.turn % len(self.lanes)]
                self.turn += 1
                if ln["primed"]:
                    ln["call"]()  # (the same C ABI call, its eight arguments converted once: Plan.bind_decode_pipelined)
                else:
                    ln["plan"].run_decode_pipelined(self.a_gp, self.a_at, ln["a_p"], None, 0, LABEL, ln["stream"])
                    ln["primed"] = True
            else:  # one pass over the CSR: state scores are accumulated once for both outputs
                self.plan.run_decode(self.d_gp.data_ptr(), self.d_at.data_ptr(), self.d_p.data_ptr(), self.d_y.data_ptr(), LABEL, 0,
                                     self.stream)

        def flush(self):
            """pipelined schedule: the labels of every lane's last batch (a Viterbi-only launch each); the next step goes to lane 0"""
            for ln in self.lanes:
                if ln["primed"]:
                    ln["plan"].flush_decode_pipelined(ln["d_y"].data_ptr(), ln["stream"])
                    ln["primed"] = False
            self.turn = 0

        def timed(self, steps, warmup, preroll_ms, schedule=None):
            """(seconds for `steps` steps, max over ranks), after `warmup` untimed ones.  Pipelined schedule: the region
            starts with an empty pipeline and ends with the flush, so it holds `steps` window passes and `steps` Viterbi
            passes (the labels of every one of its batches)."""
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < preroll_ms:
                for _ in range(20):
                    self.step(schedule)
                torch.cuda.synchronize(dev)
            for _ in range(warmup):
                self.step(schedule)
            self.flush()
            barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(schedule)
            self.issue_s = (time.perf_counter() - t0) / max(steps, 1)  # what the HOST spent per step (enqueue only, nothing waited for)
            self.flush()
            torch.cuda.synchronize(dev)
            barrier()
            elapsed = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            return elapsed

        def timed_regions(self, steps, warmup, preroll_ms, min_ms, schedule=None):
            """The K-step region above, repeated until `min_ms` have been measured (a 20-step region of this workload is
            0.6 ms: too short to mean much on its own).  Every region is exactly `steps` steps between a barrier + device
            synchronisation on both sides and starts from empty pipelines; every rank runs the same number of regions
            (decided from the first region's max-over-ranks time).  Returns the list of region times (seconds)."""
            first = self.timed(steps, warmup, preroll_ms, schedule)
            more = int(min(200, max(0, np.ceil(min_ms * 1e-3 / max(first, 1e-9)) - 1)))
            return [first] + [self.timed(steps, 0, 0.0, schedule) for _ in range(more)]

    def all_sum(v):
        if dist is None:
            return int(v)
        g = torch.tensor([int(v)], dtype=torch.int64, device=red_dev)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return int(g.item())

    # ---- workload: every rank generates its own batch (same model, different contigs) for the weak-scaling value
    if args.workload == "C1":
        # BASELINE.json configs[0]: ONE 50-gene contig on the pretrained weights (tests/golden/model.pkl), SURVEY.md 8d's
        # domain law; a step of it is one launch of one window tile + one Viterbi workgroup: launch-bound by construction
        from benchkit import latency as _lat

        c1_model = nat.Model.from_lcrf(_lat.real_blob())
        cptr, gptr, attr = _lat.c1_batch(50, c1_model.num_attrs)
        wl = dict(name="C1", w=c1_model.state_weights()[0], trans=c1_model.trans_weights()[0], contig_ptr=cptr, gene_ptr=gptr, attr_id=attr,
                  A=c1_model.num_attrs)
    else:
        wl = synth.workload(args.workload, seed=synth.SEED, law=args.synth)
    base = dict(wl)  # rank 0's batch = the batch BASELINE.json names; C4 partitions THIS one
    if rank > 0 and args.workload != "C1":
        rng = np.random.default_rng(synth.SEED + rank)
        lengths = np.diff(wl["contig_ptr"]).astype(np.int64)
        rng.shuffle(lengths)
        hot = np.argsort(wl["w"][:, 1] - wl["w"][:, 0])[-200:]
        cptr, gptr, attr = synth.synth_contigs(rng, lengths, wl["A"], planted=0.01, hot_attrs=hot)
        wl.update(contig_ptr=cptr, gene_ptr=gptr, attr_id=attr)
    model = c1_model if args.workload == "C1" else nat.Model.from_tables(wl["w"], wl["trans"])
    def lanes_for(n_genes_batch):
        if args.schedule != "pipelined" or args.windowed_only:
            return 1
        return max(1, args.streams)

    n_lanes = lanes_for(int(wl["contig_ptr"][-1]))
    res = Resident(model, wl["contig_ptr"], wl["gene_ptr"], wl["attr_id"], lanes=n_lanes)
    n_genes, nnz = res.n_genes, res.nnz

    for ln in res.lanes:
        ln["plan"].viterbi_stats(reset=True)
    regions = res.timed_regions(args.steps, args.warmup, args.preroll_ms, args.min_region_ms)
    elapsed = float(np.median(regions))
    vstats = {}
    if not args.windowed_only:  # what the Viterbi side met in those regions (crf_vd_short.hpp: the exactness margin)
        for ln in res.lanes:
            for k, v in ln["plan"].viterbi_stats(reset=True).items():
                vstats[k] = vstats.get(k, 0) + v
        vstats["batches"] = args.steps * len(regions) + args.warmup
    total_genes = all_sum(n_genes)
    pipelined = args.schedule == "pipelined" and not args.windowed_only
    # the other schedule next to the headline (a quarter of the steps)
    two_launch_ms = one_stream_ms = None
    if pipelined:
        st2 = max(args.steps // 4, 50)  # (cheap: 50 steps are 2 ms)
        two_launch_ms = res.timed(st2, min(args.warmup, 20), 0.0, schedule="two-launch") / st2 * 1e3
        if len(res.lanes) > 1:  # the same pipelined schedule on ONE stream
            keep, res.lanes = res.lanes, res.lanes[:1]
            one_stream_ms = res.timed(st2, min(args.warmup, 20), 0.0) / st2 * 1e3
            res.lanes = keep

    # ---- dominant kernel: average launch duration by HIP events on the launch stream
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
        res.plan.time_windowed(res.d_gp.data_ptr(), res.d_at.data_ptr(), res.d_p.data_ptr(), LABEL, res.stream, warmup=0, iters=100)
    kern_ms = float(np.median([res.plan.time_windowed(res.d_gp.data_ptr(), res.d_at.data_ptr(), res.d_p.data_ptr(), LABEL, res.stream, warmup=3,
                                                      iters=args.kernel_iters) for _ in range(3)]))
    alg_bytes = _alg_bytes(n_genes, nnz, res.n_contigs)
    # pipelined schedule: the launch that IS the step -- window tiles + Viterbi workgroups (crf_decode_pipelined); its
    # algorithmic bytes are the window kernel's plus one label byte per gene (the score differences the tiles leave for
    # the Viterbi workgroups of the next launch, 8 B written + 8 B read per gene, are the implementation's own traffic)
    pipe_ms = pipe_alg = None
    # (a batch with a contig longer than one 2 048-gene scan block takes the general whole-contig kernels: separate launches
    # behind the same pipelined call, and the window kernel stays the dominant one)
    one_launch = pipelined and int(np.diff(wl["contig_ptr"]).max(initial=0)) <= 2048
    if one_launch:
        # (pre-rolled like the timed regions -- after short regions with a wait each, the device is not at its clocks: the driver's
        # --steps 20 run measured 33.7 us here where the default run measures 31.9 -- and the median of three series)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
            res.plan.time_decode_pipelined(res.d_gp.data_ptr(), res.d_at.data_ptr(), res.d_p.data_ptr(), res.d_y.data_ptr(), LABEL, res.stream,
                                           warmup=0, iters=100)
        pipe_ms = float(np.median([res.plan.time_decode_pipelined(res.d_gp.data_ptr(), res.d_at.data_ptr(), res.d_p.data_ptr(),
                                                                  res.d_y.data_ptr(), LABEL, res.stream, warmup=3, iters=args.kernel_iters)
                                   for _ in range(3)]))
        pipe_alg = alg_bytes + n_genes
    # one launch ALONE on an idle device, between two events on its stream (adds the dispatch latency of the launch; detail file)
    isolated_us = None
    if one_launch and "torch_stream" in res.lanes[0]:
        ln = res.lanes[0]
        res.flush()
        ln["plan"].run_decode_pipelined(res.a_gp, res.a_at, ln["a_p"], None, 0, LABEL, ln["stream"])
        ts = []
        for _ in range(30):
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ln["torch_stream"])
            ln["plan"].run_decode_pipelined(res.a_gp, res.a_at, ln["a_p"], ln["plan"], ln["a_y"], LABEL, ln["stream"])
            e1.record(ln["torch_stream"])
            torch.cuda.synchronize(dev)
            ts.append(e0.elapsed_time(e1))
        ln["primed"] = True
        res.flush()
        isolated_us = float(np.median(ts[5:])) * 1e3
    # ... and what a launch lasts under the schedule that was timed: with two decode streams two launches are in flight,
    # each of them longer than alone, while a batch leaves every ms_per_step.  HIP events on each lane's own stream, one
    # pair per launch (the first event completes when the lane's previous launch has).
    inflight_ms = None
    if one_launch and len(res.lanes) > 1 and all("torch_stream" in ln for ln in res.lanes):
        n_ev = 200
        evs = []
        for _ in range(20):
            res.step()
        for i in range(n_ev):
            ln = res.lanes[res.turn % len(res.lanes)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ln["torch_stream"])
            res.step()
            e1.record(ln["torch_stream"])
            evs.append((e0, e1))
        res.flush()
        torch.cuda.synchronize(dev)
        inflight_ms = float(np.mean([a.elapsed_time(b) for a, b in evs[len(res.lanes):]]))
    # PMC figures of this kernel on this workload, from the committed profile of the same command
    # (tools/profile.sh -> tools/pmc_to_json.py); null when there is none
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    pmc_note = None
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path)).get(args.workload, {}) or {}
        except Exception:
            pmc = {}
        # counters of ANOTHER build of the kernel say nothing about this one: the profile records the hash of the kernel
        # source it was taken with (tools/pmc_to_json.py)
        if pmc and pmc.get("kernel_source_sha16") != _kernel_source_sha16():
            pmc_note = (f"profiles/pmc_traffic.json was taken with kernel source {pmc.get('kernel_source_sha16')}, this is "
                        f"{_kernel_source_sha16()}: counter figures dropped")
            pmc = {}
        if pmc and args.synth != "8d":
            pmc_note = "profiles/pmc_traffic.json was taken on the default weight law: counter figures dropped"
            pmc = {}
    pmc_pipe = {}
    if pipelined and os.path.exists(pmc_path):
        try:
            pmc_pipe = json.load(open(pmc_path)).get(args.workload + ":pipelined", {}) or {}
        except Exception:
            pmc_pipe = {}
        if pmc_pipe.get("kernel_source_sha16") != _kernel_source_sha16() or args.synth != "8d":
            pmc_pipe = {}
    traffic = pmc.get("hbm_bytes_per_launch")
    valu_insts = pmc.get("SQ_INSTS_VALU")  # wave-level VALU instructions of one launch
    valu_frac = (valu_insts * 4.0 / (kern_ms * 1e-3 * SIMDS * SCLK_HZ)) if valu_insts else None

    out = {
        "metric": "genes/sec CRF decode (windowed fwd-bwd marginals" + (")" if args.windowed_only else " + Viterbi)"),
        "value": total_genes * args.steps / elapsed,
        "unit": "genes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_single_region": regions[0] / args.steps * 1e3,
        # the host's share: wall time of the enqueue loop per step (last region; no wait inside).  A batch whose step is no longer than
        # this is bound by the launch path (HIP runtime ~3 us + this library ~1 us + ctypes), not by the GPU
        "host_issue_us_per_step": res.issue_s * 1e6,
        "timed_regions": {"count": len(regions), "steps_each": args.steps, "min_total_ms": args.min_region_ms,
                          "ms_per_step_min": min(regions) / args.steps * 1e3, "ms_per_step_max": max(regions) / args.steps * 1e3,
                          "note": "every region = exactly `steps` steps between barrier + device synchronisation on both sides, "
                                  "pipelines empty at its start and flushed at its end; ms_per_step / value = the MEDIAN region, "
                                  "ms_per_step_single_region = the first one"},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"C1: ONE contig of {n_genes} genes, {nnz} domain hits, on GECCO's pretrained CRF weights (tests/golden/model.pkl: "
                         f"{wl['A']} attributes, 2 labels), SURVEY.md 8d's domain law, window 20 step 1, pad -- BASELINE.json configs[0]; a "
                         "step is one launch (one window tile + one Viterbi workgroup): launch-bound, see `latency` for what a call costs")
                        if args.workload == "C1" else
                        f"{args.workload}: {res.n_contigs} contigs, {n_genes} genes, {nnz} domain hits per GPU; "
                        f"A=35000 synthetic 2-label model, window 20 step 1, pad.  " + (
                            "Side point, not SURVEY.md 8d's law: weights ~ Laplace(-0.4, 1.7) and the Zipf head (ids < A/50) forced "
                            "negative, so that most genes lean to label '0' as under the embedded model (mean w1-w0 = -0.76)"
                            if args.synth == "genome" else "Weights exactly as SURVEY.md 8d: Laplace(0, 1.7) clipped to [-6.3, 12.7]"),
            "synth_law": args.synth,
            "genes_per_gpu": n_genes,
            "viterbi_in_step": not args.windowed_only,
            "schedule": ("pipelined over batches: one launch = window tiles of a batch + Viterbi workgroups of the batch its decode "
                         "stream scored before (gecco_crf_plan_run_decode_pipelined); "
                         + (f"batches alternate between {n_lanes} independent decode streams (a plan and a HIP stream each), so "
                            f"that the tail of one launch overlaps the head of the next; " if n_lanes > 1 else "")
                         + "the timed region starts with empty pipelines and ends with their flushes: K window passes + K Viterbi "
                           f"passes in K + {n_lanes} launches" if pipelined else
                         "two launches per batch (gecco_crf_plan_run_decode)" if not args.windowed_only else "window kernel only"),
            "device_preroll_ms": args.preroll_ms,
            "sharding": "independent contig batches per rank, no collective",
        },
        "roofline": {
            **bline.roofline_record(res.plan.kernel_name, alg_bytes, kern_ms * 1e3, pmc=pmc, rocprof=pmc, step_us=None),
            "traffic_source": pmc.get("source") or pmc_note,
            "kernel_ms": kern_ms,
            # the same against the issue rate this GPU SUSTAINS on fp64 (tools/ubench/valu_rates.hip, four waves per
            # SIMD of independent chains: 4.6-4.8 "2.4 GHz cycles" per wave instruction for v_add/v_mul/v_fma_f64 and the
            # DPP moves, i.e. ~2.05 GHz effective under this load) instead of the nominal 4 cycles at 2.4 GHz
            "valu_frac_at_sustained_rate": (valu_frac * VALU_SUSTAINED_CYCLES / 4.0) if valu_frac else None,
            "ratio_form_fallback_frac": _ratio_form_fallback_fraction(wl, res.plan.num_tiles, _tile_out(res.plan, n_genes))
            if args.workload != "Cinf" else None,
        },
    }
    if vstats:
        nb = max(1, vstats.pop("batches"))
        out["viterbi_exactness"] = {
            **{k + "_per_batch": v / nb for k, v in vstats.items()},
            "note": "difference-form Viterbi: decisions within the coarse margin of a threshold (candidates), decisions inside the "
                    "margin (4 r + 4) ulp(M) in which the labels are not provably CRFsuite's, and the contigs / genes decoded again "
                    "with CRFsuite's own delta recursion because of them (DESIGN.md: exactness of V); averages per decoded batch",
        }
    if pipelined:
        out["two_launch_ms_per_step"] = two_launch_ms
        out["one_stream_ms_per_step"] = one_stream_ms
    if one_launch:
        # the step IS one launch of crf_decode_pipelined: that is the dominant kernel; the window kernel on its own (plain
        # launches, back to back) stays in the line as `roofline_window_kernel`
        out["roofline_window_kernel"] = out["roofline"]
        vi = pmc_pipe.get("SQ_INSTS_VALU")
        out["roofline"] = {
            **bline.roofline_record(pmc_pipe.get("kernel") or "crf_decode_pipelined", pipe_alg, pipe_ms * 1e3, kernel_us_isolated=isolated_us,
                                    kernel_us_in_flight=(inflight_ms * 1e3) if inflight_ms else None, launches_in_flight=n_lanes,
                                    pmc=pmc_pipe, rocprof=pmc_pipe, step_us=out["ms_per_step"] * 1e3),
            "traffic_source": pmc_pipe.get("source") or "no counter profile of this kernel source in profiles/pmc_traffic.json",
            "traffic_note": "counter traffic holds what the algorithmic bytes leave out: the hand-over between launches, partial-line "
                            "writes of the write-through stores and the tiles' halo; no array is read twice (DESIGN.md 6)",
            "kernel_ms": pipe_ms,
            "timing_note": "kernel_us: HIP events around back-to-back launches on ONE stream / launches = the interval at which "
                           "launches complete (the head of a launch overlaps the tail of the one before); kernel_us_isolated: one "
                           "launch on an idle device between two events (adds the dispatch latency); kernel_us_in_flight: the same "
                           "launch under the timed schedule (two decode streams: two launches overlap, each lasts longer, a batch "
                           "leaves every ms_per_step); kernel_us_rocprof: average begin-to-end duration in the committed "
                           "rocprofv3 --kernel-trace --stats of `bench.py --streams 1` (profiles/INDEX.md)",
            # the same at the rate batches leave the decode streams (launches overlap): how close the STEP is to the fp64
            # issue bound; x 4.7 / 4 for the rate the chip sustains on fp64 (tools/ubench/valu_rates.hip)
            "valu_frac_of_step": (vi * 4.0 / (out["ms_per_step"] * 1e-3 * SIMDS * SCLK_HZ)) if vi else None,
            "valu_frac_of_step_at_sustained_rate": (vi * VALU_SUSTAINED_CYCLES / (out["ms_per_step"] * 1e-3 * SIMDS * SCLK_HZ)) if vi else None,
        }
    if dist is not None:
        # who ran where: the process group that carried the timing barrier, and every rank's device
        devs = [None] * world
        dist.all_gather_object(devs, {"rank": rank, "device": local_rank, "name": torch.cuda.get_device_name(local_rank)})
        out["dist"] = {"backend": backend, "world_size": dist.get_world_size(), "ranks": devs}

    # ---- C4: the ONE base batch partitioned over the ranks (strong scaling, BASELINE.json configs[3])
    if world > 1:
        lengths = np.diff(base["contig_ptr"]).astype(np.int64)
        mine = sharding.partition_contigs(lengths, world)[rank]
        cptr, gptr, attr, _ = sharding.extract_shard(base["contig_ptr"], base["gene_ptr"], base["attr_id"], mine)
        shard = Resident(model, cptr, gptr, attr, lanes=lanes_for(int(cptr[-1])))
        # (pre-rolled and taken as the median region like the headline: a shard's step is a few microseconds, and a device that
        # has idled while the shard was being built needs ~10 ms of work to be back at its clocks)
        el = float(np.median(shard.timed_regions(args.steps, args.warmup, args.preroll_ms, args.min_region_ms)))
        tot = all_sum(shard.n_genes)
        out["strong_scaling"] = {
            "config": f"C4: the {args.workload} batch ({tot} genes) greedy-partitioned by gene count over {world} devices "
                      f"(gecco_amd.sharding.partition_contigs), no collective",
            "value": tot * args.steps / el, "unit": "genes/s", "ms_per_step": el / args.steps * 1e3,
            "genes_on_rank0": shard.n_genes, "scaling": "strong",
            # what the launch law of the window kernel (T = 6.5 us + 6.35 us per 1000 workgroups, measured at
            # N = 1) plus a launch-bound Viterbi kernel (~5 us + its share of the 11 us at full size) predicts for a shard
            # (pipelined schedule: ONE launch of tiles + ~genes / 2000 Viterbi workgroups on the same law; a shard's launch
            # leaves most of the chip idle, so two decode streams run two of them side by side)
            "predicted_ms_per_step": ((6.5 + 6.35 * (shard.plan.num_tiles + shard.n_genes / 2000.0) / 1000.0 + 1.5) / min(n_lanes, 2)
                                      if pipelined else
                                      (6.5 + 6.35 * shard.plan.num_tiles / 1000.0 + 5.0 + 6.0 * shard.n_genes / 2.0e6)) * 1e-3,
            "speedup_vs_one_device": (out["ms_per_step"] / (el / args.steps * 1e3)) if out.get("ms_per_step") else None,
        }

    # ---- the HBM-resident point: C3 sits in the 256 MiB Infinity Cache across repeated steps; 2e8 genes do not
    if rank == 0 and world == 1 and not args.no_past_l3 and args.workload == "C3":
        try:
            big = synth.workload("Cinf", seed=synth.SEED)
            rb = Resident(model, big["contig_ptr"], big["gene_ptr"], big["attr_id"])
            ms = rb.plan.time_windowed(rb.d_gp.data_ptr(), rb.d_at.data_ptr(), rb.d_p.data_ptr(), LABEL, rb.stream, warmup=2, iters=10)
            ab = _alg_bytes(rb.n_genes, rb.nnz, rb.n_contigs)
            out["roofline_past_l3"] = {
                "workload": f"Cinf: {rb.n_genes} genes (100 x C3), {ab / 1e9:.2f} GB algorithmic per launch: past the 256 MiB Infinity Cache",
                "kernel_ms": ms, "achieved": ab / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "genes_per_s": rb.n_genes / (ms * 1e-3),
            }
            del rb, big
            torch.cuda.empty_cache()
        except Exception as err:  # a smaller device, a busy box: the headline numbers do not depend on this point
            out["roofline_past_l3"] = {"error": str(err)}

    # ---- the strong-scaling ceiling before the hardware is there: ONE shard of the 8-way partition of this batch
    # (BASELINE.json configs[3], C4), decoded on this device.  A step of a shard is launch-bound.
    if rank == 0 and world == 1 and args.workload == "C3" and not args.no_c4:
        lengths = np.diff(base["contig_ptr"]).astype(np.int64)
        mine = sharding.partition_contigs(lengths, 8)[0]
        cptr, gptr, attr, _ = sharding.extract_shard(base["contig_ptr"], base["gene_ptr"], base["attr_id"], mine)
        sh = Resident(model, cptr, gptr, attr, lanes=lanes_for(int(cptr[-1])))
        el = float(np.median(sh.timed_regions(args.steps, min(args.warmup, 50), args.preroll_ms, args.min_region_ms)))  # (pre-rolled, median region)
        wms = sh.plan.time_windowed(sh.d_gp.data_ptr(), sh.d_at.data_ptr(), sh.d_p.data_ptr(), LABEL, sh.stream, warmup=3, iters=50)
        out["c4_shard"] = {"genes": sh.n_genes, "workgroups": sh.plan.num_tiles, "c4_shard_ms": el / args.steps * 1e3,
                           "windowed_ms": wms, "host_issue_us_per_step": sh.issue_s * 1e6, "decode_streams": len(sh.lanes),
                           "note": "rank 0's shard of the 8-way greedy partition (sharding.partition_contigs) on ONE device: "
                                   "8 devices cannot decode the batch faster than this per step"}
        out["c4_shard_ms"] = out["c4_shard"]["c4_shard_ms"]
        del sh

    # ---- the same kernel on the OTHER weight law of gecco_amd.synth (default run: 'genome', weights shifted so that most genes
    # lean to label 0 as under GECCO's embedded model)
    other = "genome" if args.synth == "8d" else "8d"
    if rank == 0 and world == 1 and args.workload == "C3" and not args.no_8d:
        w8 = synth.workload("C3", seed=synth.SEED, law=other)
        m8 = nat.Model.from_tables(w8["w"], w8["trans"])
        r8 = Resident(m8, w8["contig_ptr"], w8["gene_ptr"], w8["attr_id"], lanes=n_lanes)
        el8 = r8.timed(min(args.steps, 200), 20, 0.0)
        ms8 = r8.plan.time_windowed(r8.d_gp.data_ptr(), r8.d_at.data_ptr(), r8.d_p.data_ptr(), LABEL, r8.stream, warmup=3, iters=50)
        ab8 = _alg_bytes(r8.n_genes, r8.nnz, r8.n_contigs)
        out["roofline_" + other] = {
            "workload": f"C3 contigs, weight law '{other}' ("
                        + ("SURVEY.md 8d to the letter" if other == "8d" else "Laplace(-0.4, 1.7), Zipf head forced negative: most genes lean to label 0")
                        + f"): {r8.n_genes} genes, {r8.nnz} domain hits",
            "kernel_ms": ms8, "achieved": ab8 / (ms8 * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": ab8 / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBPS, "ms_per_step": el8 / min(args.steps, 200) * 1e3,
            "ratio_form_fallback_frac": _ratio_form_fallback_fraction(w8, r8.plan.num_tiles, _tile_out(r8.plan, r8.n_genes)),
        }
        if not args.no_levels:
            # the cluster-call levels on this law too: 8d's law puts nine genes in ten into a cluster (a degenerate refiner
            # input, and "rows + their probabilities" is then a download of nearly everything); this one calls few
            try:
                from benchkit import levels as _lv2

                out["levels_" + other + "_law"] = _lv2.cluster_levels_for(m8, w8, devices=(local_rank,))
            except Exception as err:
                out["levels_" + other + "_law"] = {"error": f"{type(err).__name__}: {err}"}
        del r8, m8, w8
        torch.cuda.empty_cache()

    # ---- SURVEY.md 8d's other levels: host buffers through the C ABI (H2D + kernel + D2H), cluster calls, table columns,
    # Gene objects -- a few iterations each; and the path `GECCO_HIP_DEVICES` users take on a multi-GPU node: ONE process,
    # one session over every visible device, host buffers in, host buffers out
    if rank == 0 and world == 1 and not args.no_levels and args.workload in ("C2", "C3"):
        from benchkit import levels

        golden = os.path.join(ROOT, "tests", "golden")
        lv = {"resident_step": {"ms": out["ms_per_step"], "genes_per_s": out["value"], "genes": n_genes,
                                "note": "the headline: inputs resident in HBM, windowed marginals + Viterbi"}}
        try:
            lv.update(levels.host_buffer_levels(model, wl, devices=(local_rank,)))
            lv["predict_tables"] = levels.tables_level(golden)
            lv["object_api"] = levels.object_level(golden)
            lv["class_levels_mode"] = ("predict_tables / object_api run ClusterCRF's default: reference-bits mode (csrc/crf_exact.hip, "
                                       "5.4x the fast window kernel's time on C3 -- invisible at these levels)")
        except Exception as err:
            lv["error"] = f"{type(err).__name__}: {err}"
        out["levels"] = lv
        nvis = torch.cuda.device_count()
        try:
            smd = {"visible_devices": nvis,
                   "one_device": levels.multi_entry_level(model, wl, (0,)),
                   # the multi-device machinery on the hardware there is: eight entries of device 0, a submitting thread each
                   "eight_entries_of_device_0": levels.multi_entry_level(model, wl, (0,) * 8)}
            if nvis > 1:
                smd["all_devices"] = levels.multi_entry_level(model, wl, tuple(range(nvis)))
                smd["speedup"] = smd["one_device"]["ms"] / smd["all_devices"]["ms"]
            else:
                smd["all_devices"] = None
                smd["note"] = "one device visible: nothing to shard over (the same call deals chunks over every listed device)"
            out["session_multi_device"] = smd
        except Exception as err:
            out["session_multi_device"] = {"error": f"{type(err).__name__}: {err}"}

    # ---- small batches: what ONE call costs at 50 ... 100 000 genes, warm and cold, at every level a caller enters the path
    # (BASELINE.json configs[0] is the first of them; gecco_amd/latency.py)
    lat_checks = None
    if rank == 0 and world == 1 and not args.no_latency and args.workload in ("C1", "C3"):
        from benchkit import latency as _lat

        try:
            block, lat_checks, _lat_model = _lat.latency_block(device=local_rank)
            block["launch_floor_note"] = ("tools/ubench/launch_floor.hip on the same box class: an EMPTY kernel launch + hipStreamSynchronize "
                                          "takes 11.9 us (12.3 us for a kernel that reads and writes pinned host memory): the floor of any "
                                          "synchronous one-shot call (profiles/r05_launch_floor.txt)")
            try:
                import subprocess

                env = dict(os.environ, GECCO_AMD_MODEL_DIR=os.path.join(ROOT, "tests", "golden"))
                cp = subprocess.run([sys.executable, "-m", "benchkit.latency", "--cold-process"], cwd=ROOT, env=env, capture_output=True,
                                    text=True, timeout=180)
                block["cold_process"] = json.loads(cp.stdout.strip().splitlines()[-1]) if cp.returncode == 0 else {"error": cp.stderr[-400:]}
            except Exception as err:
                block["cold_process"] = {"error": f"{type(err).__name__}: {err}"}
            out["latency"] = block
        except Exception as err:
            out["latency"] = {"error": f"{type(err).__name__}: {err}"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (C restatement of the CRFsuite tagger driven window by
        # window like the reference), 1 thread, on a bounded sample of the same workload.
        from oracle import crf_oracle as orc

        nc = min(len(wl["contig_ptr"]) - 1, 10000)
        cp = wl["contig_ptr"][: nc + 1]
        ng = int(cp[-1])
        # (a batch the port finishes in microseconds -- C1 -- is repeated until a fifth of a second has been measured)
        reps_cpu, dt_win, dt_vit = 0, 0.0, 0.0
        while reps_cpu == 0 or (dt_win + dt_vit < 0.2 and reps_cpu < 100000):
            t0 = time.perf_counter()
            p_ref = orc.windowed_marginals(wl["w"], wl["trans"], cp, wl["gene_ptr"][: ng + 1], wl["attr_id"], W, STEP, LABEL, True)
            dt_win += time.perf_counter() - t0
            t0 = time.perf_counter()
            y_ref, _ = orc.viterbi(wl["w"], wl["trans"], cp, wl["gene_ptr"][: ng + 1], wl["attr_id"])
            dt_vit += time.perf_counter() - t0
            reps_cpu += 1
        dt_win /= reps_cpu
        dt_vit /= reps_cpu
        dt = dt_win + (0.0 if args.windowed_only else dt_vit)
        res.d_y.fill_(7)  # (the parity check below reads the labels this step delivers, not older ones)
        res.step()
        res.flush()  # pipelined schedule: the labels of that batch come with the next call
        torch.cuda.synchronize(dev)
        got = res.d_p[:ng].cpu().numpy()
        out["cpu_baseline"] = {
            "value": ng / dt,
            "unit": "genes/s",
            "cores": 1,
            "kind": "port",
            "sample": f"first {nc} contigs ({ng} genes) of the same workload" + (f", mean of {reps_cpu} repetitions" if reps_cpu > 1 else "")
                      + f": windowed marginals {dt_win:.3g} s" + ("" if args.windowed_only else f" + Viterbi {dt_vit:.3g} s")
                      + ", C oracle driven window by window like the reference",
        }
        # SURVEY.md 8d also asks for the same restatement on all host cores (contigs over threads)
        ncpu, cpu_note = _usable_host_threads()
        best = None
        for _ in range(2):  # first threaded pass wakes the cores up
            t0 = time.perf_counter()
            orc.windowed_marginals_mt(wl["w"], wl["trans"], cp, wl["gene_ptr"][: ng + 1], wl["attr_id"], W, STEP, LABEL, True, threads=ncpu)
            if not args.windowed_only:
                orc.viterbi_mt(wl["w"], wl["trans"], cp, wl["gene_ptr"][: ng + 1], wl["attr_id"], threads=ncpu)
            d = time.perf_counter() - t0
            best = d if best is None else min(best, d)
        out["cpu_baseline_all_cores"] = {"value": ng / best, "unit": "genes/s", "cores": ncpu, "kind": "port",
                                         "speedup_over_one_thread": (ng / best) / (ng / dt),
                                         "sample": f"same sample, OpenMP over ranges of 8 contigs inside the C oracle, best of 2; {cpu_note}"}
        try:
            from benchkit import levels as _lv

            golden_tables = _lv.golden_table_identity(os.path.join(ROOT, "tests", "golden"))
            golden_tables["reference_bits_mode"] = _lv.golden_table_identity(os.path.join(ROOT, "tests", "golden"), reference_bits=True)
        except Exception as err:
            golden_tables = {"error": f"{type(err).__name__}: {err}"}
        out["parity"] = {
            "golden_tables": golden_tables,
            "max_abs_dp_vs_oracle": float(np.abs(got - p_ref).max()),
            "cluster_call_mismatches": int(((got > 0.8) != (p_ref > 0.8)).sum()),
            "genes_checked": ng,
        }
        if not args.windowed_only:
            out["parity"]["viterbi_label_mismatches"] = int((res.d_y[:ng].cpu().numpy() != y_ref.astype(np.int8)).sum())
        # what "reference-bits mode" delivers against the oracle run with the HOST'S libm exp (what CRFsuite calls): genes whose
        # probability differs in any bit, and by how many ulps at most -- next to the same count for the fast kernels
        if model.num_labels == 2:
            try:
                ses_rb = nat.Session(model, [local_rank])
                ses_rb.set_reference_bits(True)
                p_bits = np.asarray(ses_rb.windowed_marginals(cp, wl["gene_ptr"][: ng + 1], wl["attr_id"], W, STEP, LABEL, True))
                out["parity"]["reference_bits_vs_libm_oracle"] = _ulp_report(p_bits, p_ref)
                out["parity"]["fast_kernels_vs_libm_oracle"] = _ulp_report(got, p_ref)
                with orc.correctly_rounded_exp():
                    p_cr = orc.windowed_marginals_mt(wl["w"], wl["trans"], cp, wl["gene_ptr"][: ng + 1], wl["attr_id"], W, STEP, LABEL, True,
                                                     threads=ncpu)
                out["parity"]["reference_bits_vs_correctly_rounded_oracle"] = _ulp_report(p_bits, p_cr)
                del ses_rb
            except Exception as err:
                out["parity"]["reference_bits_vs_libm_oracle"] = {"error": f"{type(err).__name__}: {err}"}
        if lat_checks:
            # the C port's time for the very inputs of the latency block, and their parity (marginals, labels, cluster rows)
            lw, lt = _lat_model.state_weights()[0], _lat_model.trans_weights()[0]
            for n, ck in lat_checks.items():
                best = None
                for _ in range(3 if n <= 10000 else 1):
                    t0 = time.perf_counter()
                    lp = orc.windowed_marginals(lw, lt, ck["cptr"], ck["gptr"], ck["attr"], W, STEP, LABEL, True)
                    d1 = time.perf_counter() - t0
                    t0 = time.perf_counter()
                    ly, _ = orc.viterbi(lw, lt, ck["cptr"], ck["gptr"], ck["attr"])
                    d2 = time.perf_counter() - t0
                    best = (d1, d2) if best is None or d1 + d2 < sum(best) else best
                ann = (np.diff(ck["gptr"]) > 0).astype(np.uint8)
                lseg = orc.segment(lp, ann, ck["cptr"], 0.8, 3, 0, True)
                out["latency"][str(n)]["cpu_port"] = {
                    "windowed_us": best[0] * 1e6, "viterbi_us": best[1] * 1e6, "cores": 1,
                    "note": "the C oracle on the same input, one thread, best of 3 (driven window by window like the reference)"}
                out["latency"][str(n)]["parity"] = {
                    "max_abs_dp_vs_oracle": float(np.abs(ck["p"] - lp).max()),
                    "viterbi_label_mismatches": int((ck["y"] != ly.astype(np.int8)).sum()),
                    "cluster_rows_identical": bool(np.array_equal(ck["seg"], lseg))}
        # the TRUE reference, only if the box happens to have it (a site install; never shipped from
        # this repository): sklearn_crfsuite's tagger in the reference's own per-window Python loop
        # (gecco/crf/__init__.py:251-256) on the embedded model, 10^4 genes, one core
        ref = _true_reference_baseline()
        if ref is not None:
            out["cpu_baseline_port"] = out["cpu_baseline"]
            out["cpu_baseline"] = ref
    if rank == 0:
        # ONE short line on stdout (the driver keeps only the tail of stdout: round 5's 29 KB line could not be parsed); the whole
        # record -- levels, latency, session_multi_device, golden tables, notes -- goes to a side file the line names
        path = bline.write_detail(bline.detail_path(ROOT, args.workload, world), out)
        print(bline.dumps(bline.compact_line(out, detail=os.path.relpath(path, ROOT) if path.startswith(ROOT) else path)), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _masked_stream(pattern, k, n):
    """hipExtStreamCreateWithCUMask: decode stream k of n on a subset of the 256 CU bits.  pattern "block": bits [256 k / n, 256 (k + 1) / n);
    "interleave": bits i with (i // 8) % n == k; "half:<bits>": stream 0 gets the first <bits> of every 32, the others the rest."""
    import ctypes

    import torch

    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    if pattern == "block":
        bits = range(256 * k // n, 256 * (k + 1) // n)
    elif pattern == "interleave":
        bits = [i for i in range(256) if (i // 8) % n == k]
    elif pattern.startswith("half:"):
        c = int(pattern.split(":")[1])
        bits = [i for i in range(256) if ((i & 31) < c) == (k == 0)]
    else:
        raise SystemExit(f"GECCO_BENCH_CU_MASK={pattern!r}?")
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    h = ctypes.c_void_p()
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    if rc:
        raise SystemExit(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return h.value


def _ulp_report(a, b):
    """Two arrays of positive doubles: how many entries differ in any bit, the largest distance in ulps, out of how many."""
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    ok = np.isfinite(a) & np.isfinite(b)
    d = np.abs(a[ok].view(np.int64) - b[ok].view(np.int64))
    return {"genes_differing": int((d != 0).sum()) + int((~ok).sum() - (np.isnan(a) & np.isnan(b)).sum()), "max_ulps": int(d.max(initial=0)),
            "genes": int(a.size)}


def _usable_host_threads():
    """Threads the all-core baseline may really use: the affinity mask, capped by the cgroup CPU quota of the box (more
    threads than the quota only get throttled)."""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    note = f"{aff} hardware threads in the affinity mask"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, int(-(-int(quota) // int(period))))
            note += f", cgroup CPU quota {cores} cores"
            aff = min(aff, cores)
    except Exception:
        pass
    return aff, note


def _kernel_source_sha16():
    import hashlib

    h = hashlib.sha256()
    for name in ("crf_kernels.hip", "crf_device.hpp", "crf_vd_short.hpp", "crf_scan.hpp"):
        with open(os.path.join(ROOT, "gecco_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _tile_out(plan, n_genes):
    """Output slots per workgroup of the plan's window kernel."""
    return max(1, plan.tile_out)


def _true_reference_baseline():
    try:
        import sklearn_crfsuite  # noqa: F401
    except Exception:
        return None
    try:
        import pickle

        from gecco_amd import synth

        golden = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(golden, "model.pkl"), "rb") as fh:
            crf = pickle.load(fh)  # needs gecco + sklearn_crfsuite importable: the reference's own class
        tagger = crf.model
        rng = np.random.default_rng(synth.SEED)
        attrs = list(tagger.attributes_)
        cptr, gptr, attr = synth.synth_contigs(rng, [200] * 50, len(attrs))
        feats = [{attrs[a]: True for a in attr[gptr[g]:gptr[g + 1]]} for g in range(int(cptr[-1]))]
        t0 = time.perf_counter()
        n_win = 0
        for c in range(len(cptr) - 1):
            seq = feats[cptr[c]:cptr[c + 1]]
            probs = np.zeros(len(seq))
            for s in range(len(seq) - W + 1):  # the reference's loop, gecco/crf/__init__.py:251-256
                marg = tagger.predict_marginals_single(seq[s:s + W])
                probs[s:s + W] = np.maximum(probs[s:s + W], [m["1"] for m in marg])
                n_win += 1
        dt = time.perf_counter() - t0
        n = int(cptr[-1])
        return {"value": n / dt, "unit": "genes/s", "cores": 1, "kind": "reference",
                "sample": f"sklearn_crfsuite on the embedded model, 50 contigs x 200 genes ({n_win} windows) in {dt:.1f} s, "
                          f"the reference's per-window Python loop (marginals only)"}
    except Exception:
        return None


if __name__ == "__main__":
    main()
