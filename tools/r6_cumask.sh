# Round 6 experiment: the two decode streams on disjoint subsets of the CUs (hipExtStreamCreateWithCUMask), C5 and C3, us per step
B="python bench.py --no-levels --no-latency --no-cpu-baseline --no-past-l3 --no-8d --no-c4"
one() { GECCO_BENCH_CU_MASK=$1 GECCO_BENCH_DETAIL=/tmp/cm.json $B $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask=${1:-none} $2', 'step %.2f us' % (d['ms_per_step']*1e3))"; }
for rep in 1 2; do
for m in "" block interleave half:16 half:12 half:20; do
  one "$m" "--workload C5"
  one "$m" ""
done; done
