# Round 6: decode streams in flight (1 ... 4) on C2, C3 and the 8-way shard of C3: step time per batch (same box)
R=$PWD; O=$R/gpurun_out/r6_streams; mkdir -p $O
for rep in 1 2; do
for s in 1 2 3 4; do
  GECCO_BENCH_DETAIL=$O/c2_s$s.json python bench.py --workload C2 --streams $s --no-levels --no-latency --no-cpu-baseline --no-past-l3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 streams $s step_us %.2f' % (d['ms_per_step']*1e3))"
  GECCO_BENCH_DETAIL=$O/c3_s$s.json python bench.py --streams $s --no-levels --no-latency --no-cpu-baseline --no-past-l3 --no-8d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 streams $s step_us %.2f shard_us %.2f' % (d['ms_per_step']*1e3, d['c4_shard_ms']*1e3))"
done; done
