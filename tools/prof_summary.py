#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite outputs (kernel trace + PMC passes) into a text table.
usage: prof_summary.py <dir-with-*.db> [kernel-substring]"""
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "crf_"
    dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))
    for db in dbs:
        con = sqlite3.connect(db)
        cur = con.cursor()
        name = os.path.relpath(db, root)
        try:
            rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        except sqlite3.Error:
            rows = []
        try:
            pmc = list(cur.execute(
                "select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                "where kernel_name like ? group by kernel_name, counter_name", (f"%{pat}%",)))
        except sqlite3.Error:
            pmc = []
        if pmc:
            for k, c, v, n in pmc:
                short = k.replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
                print(f"{name}\tPMC\t{short}\t{c}\t{v:.1f}\tn={n}")
        elif rows:
            for r in rows:
                if r[4] >= 0.5:
                    print(f"{name}\tKERNEL\t{r[0][:110]}\tcalls={r[1]}\ttotal_us={r[2]:.1f}\tavg_us={r[3]:.3f}\tpct={r[4]:.2f}")


if __name__ == "__main__":
    main()
