#!/usr/bin/env python3
"""Timeline of ONE call out of a rocprofv3 --kernel-trace --memory-copy-trace run (rocpd sqlite): every kernel and copy
with start / end relative to the first event of the chosen call.  usage: timeline.py <results.db> [call-index-from-end]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
kt = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
mt = [t for t in tabs if t.startswith("memory_copies")] or [t for t in tabs if "memory_copy" in t]
for t in kt[:1]:
    cols = [c[1] for c in con.execute(f"pragma table_info({t})")]
    name = "name" if "name" in cols else "kernel_name"
    for n, s, e in con.execute(f"select {name}, start, end from {t}"):
        ev.append((s, e, "K " + n.split("(")[0].split("::")[-1][:40]))
for t in mt[:1]:
    cols = [c[1] for c in con.execute(f"pragma table_info({t})")]
    size = "size" if "size" in cols else cols[-1]
    name = "name" if "name" in cols else cols[0]
    for n, s, e, b in con.execute(f"select {name}, start, end, {size} from {t}"):
        ev.append((s, e, f"C {str(n)[:28]} {b / 1e6:.2f} MB"))
ev.sort()
# calls are separated by idle gaps > 1 ms
calls, cur = [], []
for x in ev:
    if cur and x[0] - max(c[1] for c in cur) > 1_000_000:
        calls.append(cur)
        cur = []
    cur.append(x)
calls.append(cur)
c = calls[-back]
t0 = c[0][0]
for s, e, what in c:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} us  {what}")
print("call length", (max(x[1] for x in c) - t0) / 1e3, "us;", len(calls), "calls in the trace")
