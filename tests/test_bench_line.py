"""The line bench.py prints must stay machine-safe: strict JSON, under 4 KB, with the contract's keys + roofline + cpu_baseline
(round 5's 29 KB line could not be parsed by the driver).  CPU only: canned records, no device."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchkit import line as bline  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _canned():
    roof = bline.roofline_record("crf_decode_pipelined<2, 8>", 37318269, 31.3, kernel_us_isolated=36.0, kernel_us_in_flight=47.0,
                                 launches_in_flight=2, pmc={"hbm_bytes_per_launch": 83952434, "SQ_INSTS_VALU": 13228144.0, "source": "s" * 900},
                                 rocprof={"kernel_us_rocprof": 31.939, "source": "profiles/r06_rocprofv3_summary.txt"}, step_us=27.2)
    return {
        "metric": "genes/sec CRF decode (windowed fwd-bwd marginals + Viterbi)", "value": 73447203763.3827, "unit": "genes/s", "n_gpus": 1,
        "steps": 20, "warmup": 5, "ms_per_step": 0.027230294654145837, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C3: " + "w" * 3000, "genes_per_gpu": 1999989, "schedule": "x" * 2000, "sharding": "independent contig batches per rank, no collective"},
        "roofline": {**roof, "traffic_note": "n" * 5000, "nan_field": float("nan")},
        "cpu_baseline": {"value": 1279889.2, "unit": "genes/s", "cores": 1, "kind": "port", "sample": "s" * 4000},
        "cpu_baseline_all_cores": {"value": 2.0e7, "cores": 16, "kind": "port", "sample": "t" * 4000},
        "parity": {"golden_tables": {"big": ["x"] * 1000}, "max_abs_dp_vs_oracle": 3.6e-15, "cluster_call_mismatches": 0, "viterbi_label_mismatches": 0,
                   "genes_checked": 1999989},
        "latency": {str(n): {"blob": "l" * 3000} for n in (50, 1000, 100000)},
        "levels": {"x": "y" * 5000}, "two_launch_ms_per_step": float("inf"), "c4_shard_ms": 0.0057,
    }


def test_compact_line_is_short_strict_json_with_contract_keys():
    line = bline.compact_line(_canned(), detail="gpurun_out/bench_detail_C3_n1.json")
    text = bline.dumps(line)
    assert len(text.encode()) < bline.MAX_BYTES, len(text)
    assert "\n" not in text
    back = json.loads(text, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    for k in CONTRACT:
        assert k in back, k
    for k in ("bound", "limiter", "kernel", "kernel_us", "kernel_us_rocprof", "algorithmic_bytes_per_launch", "achieved", "peak", "unit", "frac",
              "frac_rocprof", "traffic", "valu_frac"):
        assert k in back["roofline"], k
    assert back["roofline"]["limiter"] == "fp64-valu"
    assert back["cpu_baseline"]["kind"] == "port" and back["cpu_baseline"]["cores"] == 1
    assert back["two_launch_ms_per_step"] is None  # Infinity -> null
    assert "latency" not in back and "levels" not in back
    assert back["detail"].endswith(".json")


def test_roofline_record_is_recomputable_from_its_own_fields():
    r = bline.roofline_record("k", 37318269, 31.94, rocprof={"kernel_us_rocprof": 31.94}, pmc={"SQ_INSTS_VALU": 13.23e6})
    assert r["frac"] == pytest.approx(37318269 / 31.94e-6 / 8e12, rel=1e-12)
    assert r["frac"] == pytest.approx(0.146, abs=5e-4)  # the judge's round-5 recomputation
    assert r["frac_rocprof"] == pytest.approx(r["frac"], rel=1e-12)
    assert r["valu_frac"] == pytest.approx(13.23e6 * 4 / (31.94e-6 * 1024 * 2.4e9), rel=1e-12)


def test_real_round5_record_compacts():
    """The 29 KB record the driver could not parse in round 5 (profiles/r05_bench_c3_driver_command.json) -> one short line."""
    path = os.path.join(ROOT, "profiles", "r05_bench_c3_driver_command.json")
    full = json.load(open(path))
    assert len(json.dumps(full)) > 20000
    text = bline.dumps(bline.compact_line(full, detail="x.json"))
    assert len(text.encode()) < bline.MAX_BYTES
    back = json.loads(text)
    assert back["value"] == pytest.approx(full["value"], rel=1e-5)
    assert back["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)


def test_detail_file_roundtrip(tmp_path):
    p = bline.write_detail(str(tmp_path / "d.json"), _canned())
    back = json.load(open(p))
    assert back["levels"]["x"].startswith("y") and back["roofline"]["nan_field"] is None
