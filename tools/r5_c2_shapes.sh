cd $GRAFT_REPO_ROOT
for t in 2 1 3 2 1; do
  GECCO_CRF_TILES_PER_WG=$t python bench.py --workload C2 --no-levels --no-cpu-baseline --no-latency --no-past-l3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('tiles_per_wg', $t, 'step_us', round(d['ms_per_step']*1e3, 3), 'pipe_kernel_us', round(d['roofline']['kernel_ms']*1e3, 3), 'win_kernel_us', round(d['roofline_window_kernel']['kernel_ms']*1e3, 3), 'one_stream', d.get('one_stream_ms_per_step'))
"
done
