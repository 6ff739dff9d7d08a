# Round 6, same box: the decode step with and without the WRITE half of the hand-over between launches (GECCO_CRF_AB_NO_HANDOVER_STORE=1:
# the tiles keep their score differences, labels are wrong), C3 and C2; + the shard / host-issue figures of the default run
R=$PWD; O=$R/gpurun_out/r6_ab; mkdir -p $O
B="python bench.py --no-levels --no-latency --no-cpu-baseline --no-past-l3 --no-8d"
pr() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open(d['detail'] if d['detail'].startswith('/') else '$R/'+d['detail']))
print('$1', 'step %.2f us' % (d['ms_per_step']*1e3), 'one_stream %.2f' % (d.get('one_stream_ms_per_step',0)*1e3), 'pipelined launch alone %.2f' % d['roofline']['kernel_us'], 'host issue %.2f' % f['host_issue_us_per_step'], 'shard %.2f (host %.2f)' % (f['c4_shard']['c4_shard_ms']*1e3, f['c4_shard']['host_issue_us_per_step']) if 'c4_shard' in f else '')"; }
for rep in 1 2 3; do
  GECCO_BENCH_DETAIL=$O/base.json $B | pr "C3 base"
  GECCO_CRF_AB_NO_HANDOVER_STORE=1 GECCO_BENCH_DETAIL=$O/nostore.json $B | pr "C3 no_handover_store"
  GECCO_BENCH_DETAIL=$O/base2.json $B --workload C2 | pr "C2 base"
  GECCO_CRF_AB_NO_HANDOVER_STORE=1 GECCO_BENCH_DETAIL=$O/nostore2.json $B --workload C2 | pr "C2 no_handover_store"
done
