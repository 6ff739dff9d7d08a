# Same-box A/B of the product build against variant builds of the window kernel
# (tools/build_variant.sh <tag> crf_kernels.hip -D...).   gpurun -- 'bash tools/ab_diag.sh "h8 h10 h12"'
set -u
VARS=${1:-"h8 h10 h12"}
R=$PWD; O=$R/gpurun_out/r4_diag; mkdir -p $O; L=$R/gecco_amd/lib
for n in $VARS; do
  GECCO_CRF_LIBRARY=$L/libgecco_crf_$n.so timeout 900 python -m pytest tests/test_gpu_windowed.py tests/test_gpu_plan.py tests/test_gpu_fullsize.py tests/test_gpu_session.py -q > $O/pytest_$n.log 2>&1
  echo "$n: $(tail -1 $O/pytest_$n.log)"; grep -E "^(FAILED|ERROR)" $O/pytest_$n.log
done
one() {  # tag lib
  local lib=$2
  GECCO_CRF_LIBRARY=$lib timeout 300 python bench.py --no-cpu-baseline --no-past-l3 --no-levels --no-8d --no-c4 --steps 2000 --min-region-ms 0 2>> $O/bench.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1', 'step %.2f' % (d['ms_per_step'] * 1e3), 'one_stream %.2f' % (d['one_stream_ms_per_step'] * 1e3), 'two_launch %.2f' % (d['two_launch_ms_per_step'] * 1e3),
      'pipe_kernel %.2f' % (d['roofline']['kernel_ms'] * 1e3), 'win_kernel %.2f' % (d['roofline_window_kernel']['kernel_ms'] * 1e3))"
}
for round in 1 2 3; do
  one base $L/libgecco_crf.so
  for n in $VARS; do one $n $L/libgecco_crf_$n.so; done
done | tee $O/ab.txt
[ "${PMC:-0}" = 1 ] || exit 0  # counters of every build: PMC=1 (minutes per build)
for t in base $VARS; do
  lib=$L/libgecco_crf.so; [ $t != base ] && lib=$L/libgecco_crf_$t.so
  GECCO_CRF_LIBRARY=$lib tools/pmc_ab.sh r4_$t --no-c4 --no-8d > /dev/null 2>&1
  echo "== $t"; grep -h "crf_windowed_l2" $R/gpurun_out/pmc_r4_$t/summary.txt | awk -F'\t' '{printf "%s=%s ", $4, $5}'; echo
done | tee $O/pmc.txt
rm -rf $R/gpurun_out/pmc_r4_*/pmc*/  # (the raw databases stay on the box: only the summaries travel back)
