R=$PWD; L=$R/gecco_amd/lib; O=$R/gpurun_out/r6_ab3; mkdir -p $O
B="python bench.py --no-levels --no-latency --no-cpu-baseline --no-past-l3 --no-8d --no-c4"
one() { GECCO_CRF_LIBRARY=$2 GECCO_BENCH_DETAIL=$O/$1.json $B $3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $3', 'step %.2f us' % (d['ms_per_step']*1e3), 'one_stream %.2f' % (d.get('one_stream_ms_per_step',0)*1e3), 'pipelined launch alone %.2f' % d['roofline']['kernel_us'])"; }
for rep in 1 2 3; do
  one base $L/libgecco_crf.so ""
  one dstate_plain $L/libgecco_crf_PLAIN.so ""
  one dstate_nt $L/libgecco_crf_NT.so ""
done
