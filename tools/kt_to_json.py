#!/usr/bin/env python3
"""Average begin-to-end duration of a kernel in a `rocprofv3 --kernel-trace --stats` run (rocpd sqlite) -> `kernel_us_rocprof` of
an entry of profiles/pmc_traffic.json (and of profiles/<tag>_pmc.json): the committed figure `bench.py` quotes next to its live
HIP-event timing (`roofline.kernel_us_rocprof`, `roofline.frac_rocprof`).
usage: kt_to_json.py <dir-with-kt db> <workload[:pipelined]> <tag> <kernel-substring> [note]"""
import glob
import json
import os
import sqlite3
import sys


def main():
    root, workload, tag, pat = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --kernel-trace --stats of `bench.py --streams 1` (launches of ONE decode stream: begin-to-end durations)"
    rows = []
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        try:
            rows += list(sqlite3.connect(db).execute("select name,total_calls,total_duration,average from top_kernels where name like ?", (f"%{pat}%",)))
        except sqlite3.Error:
            pass
    if not rows:
        raise SystemExit(f"no kernel matching {pat!r} under {root}")
    name, calls, total, avg = max(rows, key=lambda r: r[1])
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in (os.path.join(repo, "profiles", "pmc_traffic.json"), os.path.join(repo, "profiles", f"{tag}_pmc.json")):
        try:
            doc = json.load(open(path))
        except Exception:
            doc = {}
        e = doc.setdefault(workload, {})
        e["kernel_us_rocprof"] = float(avg)  # (the view reports microseconds: tools/prof_summary.py prints the same column)
        e["kernel_us_rocprof_calls"] = int(calls)
        e["kernel_us_rocprof_source"] = f"profiles/{tag}_rocprofv3_summary.txt <- {os.path.relpath(os.path.abspath(root), repo)}: {note}"
        json.dump(doc, open(path, "w"), indent=1)
    print(workload, name[:80], "calls", calls, "avg_us", float(avg))


if __name__ == "__main__":
    main()
