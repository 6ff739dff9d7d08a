set -u
R=$PWD; O=$R/gpurun_out/r5_c5kt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for st in 1 2; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt$st -o kt -- python bench.py --workload C5 --no-past-l3 --no-levels --no-latency --no-cpu-baseline --steps 200 --warmup 20 --min-region-ms 0 --streams $st > $O/kt$st.log 2>&1
python tools/prof_summary.py $O/kt$st "" | cut -c1-200 > $O/kt$st.txt
done
# a timeline of ~3 steps under two streams
db=$(ls $O/kt2/*/*.db $O/kt2/*.db 2>/dev/null | head -1)
python - "$db" > $O/timeline2.txt <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")][0]
cols = [c[1] for c in con.execute(f"pragma table_info({kt})")]
name = "name" if "name" in cols else "kernel_name"
extra = ", stream_id" if "stream_id" in cols else (", queue_id" if "queue_id" in cols else "")
rows = list(con.execute(f"select {name}, start, end{extra} from {kt} order by start"))
mid = len(rows) * 2 // 3
t0 = rows[mid][1]
for r in rows[mid:mid + 40]:
    n = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1]
    print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-t0)/1e3:9.1f} ({(r[2]-r[1])/1e3:6.1f}) q{r[3] if extra else ''} {n}")
PY
rm -rf $O/kt1/*/ $O/kt2/*/ 2>/dev/null
cat $O/kt2.txt; cat $O/timeline2.txt
