# Round-6 profile set, ONE gpurun call:  gpurun --timeout 3000 -- 'bash tools/r6_profiles.sh'
# Everything lands under gpurun_out/r6_final/ (summaries + the small rocpd databases of the C3 passes); tools/collect_r6.sh copies
# the judged files into profiles/.
set -u
R=$PWD; O=$R/gpurun_out/r6_final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
kt() {  # <name> <command...>: rocprofv3 --kernel-trace --stats of a command -> per-kernel calls / average duration
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o kt -- "$@" > $O/kt_$name.log 2>&1
  python tools/prof_summary.py $O/kt_$name "" | cut -c1-260 > $O/kt_$name.txt
  rm -rf $O/kt_$name
}
bench() {  # <name> <bench args...>: the printed line -> <name>.json, the full record -> <name>_detail.json
  local name=$1; shift
  GECCO_BENCH_DETAIL=$O/${name}_detail.json timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err
}
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 1500 tools/profile.sh r6_final --no-latency > $O/profile.log 2>&1
C5="python bench.py --workload C5 --windowed-only --steps 3 --warmup 1 --kernel-iters 3 --preroll-ms 0 --no-cpu-baseline --no-levels --no-latency --no-past-l3 --min-region-ms 0"
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/c5win/pmc2 -o pmc2 -- $C5 > $O/c5win_pmc2.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/c5win/pmc3 -o pmc3 -- $C5 > $O/c5win_pmc3.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/c5win/pmc1 -o pmc1 -- $C5 > $O/c5win_pmc1.log 2>&1
# the hand-over A/B (item 6 of the round-5 review): counters of the pipelined launch with the tiles' score-difference stores off
S2="--no-cpu-baseline --no-past-l3 --no-c4 --no-8d --no-levels --no-latency --streams 1 --steps 4 --warmup 1 --kernel-iters 3 --preroll-ms 0 --min-region-ms 0"
GECCO_CRF_AB_NO_HANDOVER_STORE=1 timeout 180 rocprofv3 --pmc FETCH_SIZE -d $O/nostore/pmc2 -o pmc2 -- python bench.py $S2 > $O/nostore_pmc2.log 2>&1
GECCO_CRF_AB_NO_HANDOVER_STORE=1 timeout 180 rocprofv3 --pmc WRITE_SIZE -d $O/nostore/pmc3 -o pmc3 -- python bench.py $S2 > $O/nostore_pmc3.log 2>&1
GECCO_CRF_AB_NO_HANDOVER_STORE=1 timeout 180 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/nostore/pmc1 -o pmc1 -- python bench.py $S2 > $O/nostore_pmc1.log 2>&1
python tools/prof_summary.py $O/nostore crf_decode > $O/nostore_summary.txt 2>&1
bash tools/r6_handover_ab.sh > $O/handover_ab.txt 2>&1
bench bench_c3_driver --gpus 1 --steps 20 --warmup 5
bench bench_c3
bench bench_c3_two_launch --schedule two-launch --no-levels --no-latency --no-cpu-baseline
bench bench_c5 --workload C5 --no-past-l3 --no-latency
bench bench_c5_driver --workload C5 --no-past-l3 --no-latency --steps 20 --warmup 5
bench bench_c2 --workload C2 --no-past-l3 --no-levels --no-latency
bench bench_c1 --workload C1
timeout 400 python tools/bench_levels.py > $O/levels.json 2> $O/levels.err
timeout 300 python tools/bench_full.py > $O/whole_contig.json 2> $O/whole_contig.err
timeout 400 python tools/bench_general.py > $O/general_l.json 2> $O/general_l.err
timeout 200 python tools/direct_sweep.py > $O/direct_sweep.json 2> $O/direct_sweep.err
GECCO_BENCH_DETAIL=$O/bench_world1_nccl_detail.json GECCO_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-past-l3 --no-levels --no-latency --no-8d --no-c4 2> $O/bench_world1_nccl.err | grep '^{' | tail -1 > $O/bench_world1_nccl.json
GECCO_BENCH_DETAIL=$O/bench_2ranks_detail.json GECCO_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline --no-past-l3 --no-latency 2> $O/bench_2ranks.err | grep '^{' | tail -1 > $O/bench_2ranks_one_device.json
kt c5 python bench.py --workload C5 --no-past-l3 --no-cpu-baseline --no-latency --steps 200 --warmup 20 --min-region-ms 0
kt whole_contig python tools/bench_full.py
kt general_l python tools/bench_general.py 3 8 16 32
kt levels python tools/bench_levels.py
bash tools/r5_latency.sh final > $O/latency_run.log 2>&1
cp $R/gpurun_out/r5_lat_final/* $O/ 2>/dev/null
timeout 120 tools/ubench/pcie_bw > $O/pcie_bw.txt 2>&1
for g in 2000 200000; do timeout 120 python tools/host_issue_probe.py $g; done > $O/host_issue.txt 2>&1
bash tools/r6_streams.sh > $O/streams.txt 2>&1
timeout 900 python tools/refbits_bench.py 2> /dev/null | grep '^{' > $O/reference_bits.jsonl
GECCO_CRF_REFERENCE_BITS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_ref -o kt -- python tools/refbits_bench.py resident > $O/kt_ref.log 2>&1
python tools/prof_summary.py $O/kt_ref "" | cut -c1-260 | head -4 > $O/kt_reference_bits.txt; rm -rf $O/kt_ref
python -m pytest tests/test_gpu_reference_bits.py -q -s -k c3 2>&1 | grep "C3, " > $O/reference_bits_vs_libm.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json; cp profiles/r6_final_pmc.json $O/ 2>/dev/null
tail -1 $O/bench_c3_driver.json | cut -c1-400
ls $O | head -100; du -sh $O
