set -u
R=$PWD; O=$R/gpurun_out/r5_ab; mkdir -p $O
python -m pytest tests/test_gpu_sequence.py tests/test_gpu_fullsize.py tests/test_gpu_session.py tests/test_gpu_plan.py -m gpu -q 2>&1 | tail -2
c5() { python bench.py --workload C5 --no-past-l3 --no-levels --no-latency --no-cpu-baseline "$@" 2>> $O/c5.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$TAG', 'step %.2f us' % (d['ms_per_step'] * 1e3), 'one_stream %.2f' % ((d.get('one_stream_ms_per_step') or 0) * 1e3), 'two_launch %.2f' % ((d.get('two_launch_ms_per_step') or 0) * 1e3), 'window kernel %.2f' % (d['roofline']['kernel_ms'] * 1e3))"; }
for r in 1 2; do
  TAG=fold_under_tiles; c5 --steps 200 --warmup 20
  TAG=driver_cmd; c5 --steps 20 --warmup 5
  TAG=fold_as_a_launch; GECCO_CRF_FOLD_UNDER_TILES=0 c5 --steps 200 --warmup 20
done | tee $O/c5_fold.txt
