# Round-5 profile set, ONE gpurun call:  gpurun --timeout 3000 -- 'bash tools/r5_profiles.sh'
# Everything lands under gpurun_out/r5_final/ (summaries only: the raw rocprofv3 databases stay on the box).
set -u
R=$PWD; O=$R/gpurun_out/r5_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
kt() {  # <name> <command...>: rocprofv3 --kernel-trace --stats of a command -> per-kernel calls / average duration
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o kt -- "$@" > $O/kt_$name.log 2>&1
  python tools/prof_summary.py $O/kt_$name "" | cut -c1-260 > $O/kt_$name.txt
  rm -rf $O/kt_$name
}
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 1200 tools/profile.sh r5_final --no-latency > $O/profile.log 2>&1
C5="python bench.py --workload C5 --windowed-only --steps 3 --warmup 1 --kernel-iters 3 --preroll-ms 0 --no-cpu-baseline --no-levels --no-latency --no-past-l3 --min-region-ms 0"
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/c5win/pmc2 -o pmc2 -- $C5 > $O/c5win_pmc2.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/c5win/pmc3 -o pmc3 -- $C5 > $O/c5win_pmc3.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/c5win/pmc1 -o pmc1 -- $C5 > $O/c5win_pmc1.log 2>&1
python tools/pmc_to_json.py $O/c5win C5 r05 crf_windowed_l2 >> $O/pmc.json 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver.json 2> $O/bench_c3_driver.err
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
timeout 300 python bench.py --schedule two-launch --no-levels --no-latency --no-cpu-baseline > $O/bench_c3_two_launch.json 2> $O/bench_c3_two_launch.err
timeout 300 python bench.py --workload C5 --no-past-l3 --no-latency > $O/bench_c5.json 2> $O/bench_c5.err
timeout 300 python bench.py --workload C5 --no-past-l3 --no-latency --steps 20 --warmup 5 > $O/bench_c5_driver.json 2> $O/bench_c5_driver.err
timeout 300 python bench.py --workload C2 --no-past-l3 --no-levels --no-latency > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --workload C1 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 400 python tools/bench_levels.py > $O/levels.json 2> $O/levels.err
timeout 300 python tools/bench_full.py > $O/whole_contig.json 2> $O/whole_contig.err
timeout 400 python tools/bench_general.py > $O/general_l.json 2> $O/general_l.err
timeout 200 python tools/direct_sweep.py > $O/direct_sweep.json 2> $O/direct_sweep.err
GECCO_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-past-l3 --no-levels --no-latency --no-8d --no-c4 2> $O/bench_world1_nccl.err | grep '^{' | tail -1 > $O/bench_world1_nccl.json
GECCO_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline --no-past-l3 --no-latency 2> $O/bench_2ranks.err | grep '^{' | tail -1 > $O/bench_2ranks_one_device.json
kt c5 python bench.py --workload C5 --no-past-l3 --no-cpu-baseline --no-latency --steps 200 --warmup 20 --min-region-ms 0
kt whole_contig python tools/bench_full.py
kt general_l python tools/bench_general.py 3 8 16 32
kt levels python tools/bench_levels.py
# latency set: launch floor, latency block, cold process, timelines of one warm C1 call (kernels + copies + HIP API)
bash tools/r5_latency.sh final > $O/latency_run.log 2>&1
cp $R/gpurun_out/r5_lat_final/* $O/ 2>/dev/null
timeout 120 tools/ubench/pcie_bw > $O/pcie_bw.txt 2>&1
# timeline of one C3 decode call through the batch driver (kernels + copies), both HIP runtimes
bash tools/r5_tl.sh > $O/tl_run.log 2>&1
cp $R/gpurun_out/r5_tl/timeline_decode_*.txt $O/ 2>/dev/null
python tools/multi_entry_probe.py 1 2 4 8 > $O/multi_entry.txt 2>&1
# reference-bits mode against the fast kernels (resident, one-shot, C1, the class's levels) + its two kernels' durations
timeout 900 python tools/refbits_bench.py 2> /dev/null | grep '^{' > $O/reference_bits.jsonl
GECCO_CRF_REFERENCE_BITS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_ref -o kt -- python tools/refbits_bench.py resident > $O/kt_ref.log 2>&1
python tools/prof_summary.py $O/kt_ref "" | cut -c1-260 | head -4 > $O/kt_reference_bits.txt; rm -rf $O/kt_ref
timeout 300 python tools/tiles_sweep.py > $O/tiles_sweep.txt 2>&1
tail -1 $O/bench_c3_driver.json | cut -c1-300
ls $O | head -80
