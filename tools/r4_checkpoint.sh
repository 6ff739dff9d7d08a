# Round 4 checkpoint, ONE gpurun call: GPU suite, same-box A/B of a variant build, bench lines (C3, C5), nccl at world size 1.
#   gpurun --timeout 2400 -- 'bash tools/r4_checkpoint.sh <outdir> [variant tag]'
set -u
TAGDIR=${1:-r4a}; VAR=${2:-}
R=$PWD; O=$R/gpurun_out/$TAGDIR; mkdir -p $O; L=$R/gecco_amd/lib
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "suite: $(tail -1 $O/pytest.log)"; grep -E "^(FAILED|ERROR)" $O/pytest.log
one() {  # tag lib
  GECCO_CRF_LIBRARY=$2 timeout 300 python bench.py --no-cpu-baseline --no-past-l3 --no-levels --no-8d --no-c4 --steps 2000 --min-region-ms 0 2>> $O/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', 'step %.2f' % (d['ms_per_step'] * 1e3), 'one_stream %.2f' % (d['one_stream_ms_per_step'] * 1e3), 'two_launch %.2f' % (d['two_launch_ms_per_step'] * 1e3),
      'pipe_kernel %.2f' % (d['roofline']['kernel_ms'] * 1e3), 'in_flight %.2f' % (d['roofline']['kernel_ms_in_flight'] * 1e3), 'win_kernel %.2f' % (d['roofline_window_kernel']['kernel_ms'] * 1e3))"
}
if [ -n "$VAR" ]; then
  for round in 1 2 3; do one base $L/libgecco_crf.so; one $VAR $L/libgecco_crf_$VAR.so; done | tee $O/ab.txt
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver.json 2> $O/bench_c3_driver.err; tail -c 600 $O/bench_c3_driver.err
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
timeout 300 python bench.py --workload C5 --no-past-l3 --no-levels > $O/bench_c5.json 2> $O/bench_c5.err
python - <<PY
import json
for f in ("bench_c3_driver", "bench_c3", "bench_c5"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "step_us %.2f single %.2f regions %d value %.3g" % (d["ms_per_step"] * 1e3, d["ms_per_step_single_region"] * 1e3, d["timed_regions"]["count"], d["value"]),
              d.get("viterbi_exactness"), d.get("parity"), {k: round(v["ms"], 3) for k, v in d.get("levels", {}).items() if isinstance(v, dict) and "ms" in v})
    except Exception as e:
        print(f, "unreadable:", e)
PY
GECCO_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-past-l3 --no-levels --no-8d --no-c4 > $O/bench_world1_nccl.json 2> $O/bench_world1_nccl.err
python -c "
import json; d = json.loads(open('$O/bench_world1_nccl.json').read().strip().splitlines()[-1]); print('world1:', d.get('dist'))" || tail -5 $O/bench_world1_nccl.err
