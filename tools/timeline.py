#!/usr/bin/env python3
"""Timeline of ONE call out of a rocprofv3 --kernel-trace --memory-copy-trace [--hip-runtime-trace] run (rocpd sqlite): every
kernel, copy and (when traced) HIP API call with start / end relative to the first event of the chosen call.
usage: timeline.py <results.db> [call-index-from-end] [gap-us separating calls, default 1000] [--schema]"""
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
con = sqlite3.connect(args[0])
back = int(args[1]) if len(args) > 1 else 2
gap_ns = int(float(args[2]) * 1e3) if len(args) > 2 else 1_000_000
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
if "--schema" in sys.argv:
    for t in tabs:
        print(t, [c[1] for c in con.execute(f"pragma table_info({t})")])
ev = []
kt = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
mt = [t for t in tabs if t.startswith("memory_copies")] or [t for t in tabs if "memory_copy" in t]
rt = [t for t in tabs if t.startswith("regions")]
for t in kt[:1]:
    cols = [c[1] for c in con.execute(f"pragma table_info({t})")]
    name = "name" if "name" in cols else "kernel_name"
    extra = ", grid_x, workgroup_x" if "grid_x" in cols and "workgroup_x" in cols else ""
    for row in con.execute(f"select {name}, start, end{extra} from {t}"):
        n, s, e = row[:3]
        g = f" [{row[3] // max(row[4], 1)} wg]" if extra else ""
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1]
        ev.append((s, e, "K " + short[:44] + g))
for t in mt[:1]:
    cols = [c[1] for c in con.execute(f"pragma table_info({t})")]
    size = "size" if "size" in cols else cols[-1]
    name = "name" if "name" in cols else cols[0]
    for n, s, e, b in con.execute(f"select {name}, start, end, {size} from {t}"):
        ev.append((s, e, f"C {str(n)[:28]} {b} B"))
api = []
for t in rt[:1]:
    cols = [c[1] for c in con.execute(f"pragma table_info({t})")]
    if {"name", "start", "end"} <= set(cols):
        for n, s, e in con.execute(f"select name, start, end from {t}"):
            api.append((s, e, "A " + str(n)[:44]))
ev.sort()
# calls are separated by idle gaps of the device
calls, cur = [], []
for x in ev:
    if cur and x[0] - max(c[1] for c in cur) > gap_ns:
        calls.append(cur)
        cur = []
    cur.append(x)
calls.append(cur)
c = calls[-back] if len(calls) >= back else calls[-1]
d0, d1 = c[0][0], max(x[1] for x in c)
# API calls that overlap the window [first device event - 100 us, last device event + 30 us]
host = [x for x in api if x[1] >= d0 - 100_000 and x[0] <= d1 + 30_000]
allev = sorted(c + host)
t0 = allev[0][0]
for s, e, what in allev:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  {what}")
print("device span of the call", (d1 - d0) / 1e3, "us;", len(calls), "calls in the trace")
