# Round 6, same box: the per-window form choice (the tile path spills `tid` once per phase for it) against the build before it
R=$PWD; L=$R/gecco_amd/lib; O=$R/gpurun_out/r6_cut; mkdir -p $O
B="python bench.py --no-levels --no-latency --no-cpu-baseline --no-past-l3 --no-8d"
one() { GECCO_CRF_LIBRARY=$2 GECCO_BENCH_DETAIL=$O/$1.json $B $3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('$O/$1.json'))
print('$1 $3', 'step %.2f us' % (d['ms_per_step']*1e3), 'one_stream %.2f' % (d.get('one_stream_ms_per_step',0)*1e3), 'pipelined launch alone %.2f' % d['roofline']['kernel_us'], 'window kernel alone %.2f' % f['roofline_window_kernel']['kernel_us'] if 'roofline_window_kernel' in f else '')"; }
for rep in 1 2 3; do
  one base $L/libgecco_crf_base.so ""
  one per_window $L/libgecco_crf.so ""
  one base $L/libgecco_crf_base.so "--workload C5"
  one per_window $L/libgecco_crf.so "--workload C5"
done
