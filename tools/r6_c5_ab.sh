# Round 6, same box: C5 with vd_replay folding and scanning its own lanes again (no 24 B per lane through memory) against the build before
R=$PWD; L=$R/gecco_amd/lib
B="python bench.py --workload C5 --no-levels --no-latency --no-cpu-baseline --no-past-l3"
one() { GECCO_CRF_LIBRARY=$2 GECCO_BENCH_DETAIL=/tmp/c5.json $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('/tmp/c5.json'))
print('$1', 'step %.2f us' % (d['ms_per_step']*1e3), 'one_stream %.2f' % (d.get('one_stream_ms_per_step',0)*1e3), 'two_launch %.2f' % (d.get('two_launch_ms_per_step',0)*1e3), 'label mismatches', f.get('parity',{}).get('viterbi_label_mismatches'))"; }
python -m pytest tests/test_gpu_sequence.py tests/test_gpu_fullsize.py tests/test_gpu_plan.py -q -x 2>&1 | tail -1
for rep in 1 2 3; do one base $L/libgecco_crf_base.so; one replay_refolds $L/libgecco_crf.so; done
python tools/bench_full.py 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:{a:round(b['ms']*1e3,1) for a,b in v.items() if isinstance(b,dict)} for k,v in d.items()})"
GECCO_CRF_LIBRARY=$L/libgecco_crf_base.so python tools/bench_full.py 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', {k:{a:round(b['ms']*1e3,1) for a,b in v.items() if isinstance(b,dict)} for k,v in d.items()})"
