"""The ONE line `bench.py` prints, and the side file that holds everything else.

The driver keeps only the tail of stdout: round 5's line had grown to 29 KB and could not be parsed.  `compact_line` picks the
contract's keys (+ `roofline`, `cpu_baseline`, `parity`) out of the full record and is checked (tests/test_bench_line.py) to
stay under `MAX_BYTES` as strict JSON; the full record goes to `detail_path()` and the line names that path.
"""
import json
import math
import os

MAX_BYTES = 4096
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E, 8 TB/s


def _num(v, digits=6):
    """Finite floats rounded to `digits` significant digits (strict JSON has no NaN / Infinity: those become null)."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, int):
        return v
    if isinstance(v, float):
        if not math.isfinite(v):
            return None
        return float(f"{v:.{digits}g}")
    return v


def _pick(src, keys, digits=6):
    return {k: _num(src.get(k), digits) for k in keys if k in src}


def roofline_record(kernel, alg_bytes, kernel_us_events, *, kernel_us_isolated=None, kernel_us_in_flight=None, launches_in_flight=1,
                    pmc=None, rocprof=None, step_us=None):
    """One reproducible roofline record of the dominant kernel.

    `achieved` / `frac` come from the LIVE figure: algorithmic bytes of one launch / `kernel_us` (HIP events on the launch stream
    around back-to-back launches on ONE stream: the interval at which launches complete).  `kernel_us_rocprof` is the average
    begin-to-end duration of the same kernel in the committed `rocprofv3 --kernel-trace --stats` summary of
    `bench.py --streams 1` (profiles/pmc_traffic.json: `kernel_us_rocprof`, with the kernel source hash it was taken on);
    `frac_rocprof` = the same bytes / that duration / peak, so a reader can recompute it from the tracked profile alone.
    `limiter`: what bounds the kernel in practice -- fp64 VALU issue -- with `valu_frac` = wave-level VALU instructions of one
    launch (PMC SQ_INSTS_VALU) x 4 cycles / (kernel_us x 1024 SIMDs x 2.4 GHz)."""
    pmc = pmc or {}
    rocprof = rocprof or {}
    us = float(kernel_us_events)
    achieved = alg_bytes / (us * 1e-6) / 1e9
    rec = {
        "bound": "hbm",
        "limiter": "fp64-valu",
        "kernel": kernel,
        "kernel_us": us,
        "kernel_us_isolated": kernel_us_isolated,
        "kernel_us_in_flight": kernel_us_in_flight,
        "launches_in_flight": launches_in_flight,
        "kernel_us_rocprof": rocprof.get("kernel_us_rocprof"),
        "algorithmic_bytes_per_launch": int(alg_bytes),
        "achieved": achieved,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        "frac_rocprof": (alg_bytes / (rocprof["kernel_us_rocprof"] * 1e-6) / 1e9 / HBM_PEAK_GBPS) if rocprof.get("kernel_us_rocprof") else None,
        "frac_of_step": (alg_bytes / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if step_us else None,
        "traffic": pmc.get("hbm_bytes_per_launch"),
        "valu_insts_per_launch": pmc.get("SQ_INSTS_VALU"),
        "valu_frac": (pmc["SQ_INSTS_VALU"] * 4.0 / (us * 1e-6 * 1024 * 2.4e9)) if pmc.get("SQ_INSTS_VALU") else None,
        "profile": rocprof.get("source") or pmc.get("source"),
    }
    return rec


_ROOF_KEYS = ("bound", "limiter", "kernel", "kernel_us", "kernel_us_isolated", "kernel_us_in_flight", "launches_in_flight", "kernel_us_rocprof",
              "algorithmic_bytes_per_launch", "achieved", "peak", "unit", "frac", "frac_rocprof", "frac_of_step", "traffic",
              "valu_insts_per_launch", "valu_frac")


def compact_line(full, detail=None):
    """The printed line: contract keys + roofline + cpu_baseline + parity, nothing else.  `full` is bench.py's whole record."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"))
    cfg = full.get("config", {})
    line["config"] = {
        "workload": str(cfg.get("workload", ""))[:260],
        "genes_per_gpu": cfg.get("genes_per_gpu"),
        "schedule": str(cfg.get("schedule", ""))[:160],
        "sharding": cfg.get("sharding"),
    }
    if "roofline" in full:
        line["roofline"] = _pick(full["roofline"], _ROOF_KEYS, 5)
        prof = full["roofline"].get("profile")
        if prof:
            line["roofline"]["profile"] = str(prof)[:160]
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {**_pick(cb, ("value", "unit", "cores", "kind"), 5), "sample": str(cb.get("sample", ""))[:200]}
        if cb.get("value"):
            line["speedup_vs_cpu_baseline"] = _num(full["value"] / cb["value"] / max(1, full.get("n_gpus", 1)), 4)
    if "cpu_baseline_all_cores" in full:
        line["cpu_baseline_all_cores"] = _pick(full["cpu_baseline_all_cores"], ("value", "cores", "kind"), 5)
    if "parity" in full:
        line["parity"] = _pick(full["parity"], ("max_abs_dp_vs_oracle", "cluster_call_mismatches", "viterbi_label_mismatches", "genes_checked",
                                                "reference_bits_vs_libm_oracle"), 3)
    for k in ("two_launch_ms_per_step", "one_stream_ms_per_step", "c4_shard_ms"):
        if full.get(k) is not None:
            line[k] = _num(full[k], 5)
    if "strong_scaling" in full:
        line["strong_scaling"] = _pick(full["strong_scaling"], ("value", "unit", "ms_per_step", "genes_on_rank0", "scaling", "speedup_vs_one_device"))
    if "dist" in full:
        line["dist"] = {"backend": full["dist"].get("backend"), "world_size": full["dist"].get("world_size")}
    line["detail"] = detail
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text.encode()) >= MAX_BYTES:  # never print a line the driver cannot keep: drop the optional blocks, longest first
        for k in ("cpu_baseline_all_cores", "strong_scaling", "dist", "two_launch_ms_per_step", "one_stream_ms_per_step"):
            line.pop(k, None)
        line["config"]["workload"] = line["config"]["workload"][:120]
        line["config"]["schedule"] = line["config"]["schedule"][:60]
    return line


def dumps(line):
    """Strict JSON (no NaN / Infinity), no spaces."""
    return json.dumps(line, allow_nan=False, separators=(",", ":"))


def detail_path(root, workload, n_gpus):
    """Where the full record goes: $GECCO_BENCH_DETAIL, else gpurun_out/ under the repository (merged back by gpurun), else /tmp."""
    p = os.environ.get("GECCO_BENCH_DETAIL")
    if p:
        return p
    name = f"bench_detail_{workload}_n{n_gpus}.json"
    for d in (os.path.join(root, "gpurun_out"), root, "/tmp"):
        try:
            os.makedirs(d, exist_ok=True)
            if os.access(d, os.W_OK):
                return os.path.join(d, name)
        except OSError:
            continue
    return os.path.join("/tmp", name)


def write_detail(path, full):
    def clean(o):
        if isinstance(o, float):
            return o if math.isfinite(o) else None
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        return o

    try:
        with open(path, "w") as fh:
            json.dump(clean(full), fh, indent=1, allow_nan=False)
        return path
    except OSError as err:
        return f"(not written: {err})"
