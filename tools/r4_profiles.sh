# Round-4 profile set, ONE gpurun call:  /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/r4_profiles.sh'
# Everything lands under gpurun_out/r4_final/ (summaries only: the raw rocprofv3 databases stay on the box).
set -u
R=$PWD; O=$R/gpurun_out/r4_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
kt() {  # <name> <command...>: rocprofv3 --kernel-trace --stats of a command -> per-kernel calls / average duration
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o kt -- "$@" > $O/kt_$name.log 2>&1
  python tools/prof_summary.py $O/kt_$name "" | cut -c1-260 > $O/kt_$name.txt
  rm -rf $O/kt_$name
}
timeout 1200 tools/profile.sh r4_final > $O/profile.log 2>&1
# counters of the window kernel on C5 (the dominant kernel of `bench.py --workload C5`): its roofline.traffic
mkdir -p $R/profiles
C5="python bench.py --workload C5 --windowed-only --steps 3 --warmup 1 --kernel-iters 3 --preroll-ms 0 --no-cpu-baseline --no-levels --no-past-l3 --min-region-ms 0"
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/c5win/pmc2 -o pmc2 -- $C5 > $O/c5win_pmc2.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/c5win/pmc3 -o pmc3 -- $C5 > $O/c5win_pmc3.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/c5win/pmc1 -o pmc1 -- $C5 > $O/c5win_pmc1.log 2>&1
python tools/pmc_to_json.py $O/c5win C5 r4_final crf_windowed_l2 >> $O/pmc.json 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver.json 2> $O/bench_c3_driver.err
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
timeout 300 python bench.py --schedule two-launch --no-levels --no-cpu-baseline > $O/bench_c3_two_launch.json 2> $O/bench_c3_two_launch.err
timeout 300 python bench.py --workload C5 --no-past-l3 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 300 python bench.py --workload C2 --no-past-l3 --no-levels > $O/bench_c2.json 2> $O/bench_c2.err
timeout 400 python tools/bench_levels.py > $O/levels.json 2> $O/levels.err
timeout 300 python tools/bench_full.py > $O/whole_contig.json 2> $O/whole_contig.err
timeout 400 python tools/bench_general.py > $O/general_l.json 2> $O/general_l.err
GECCO_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-past-l3 --no-levels --no-8d --no-c4 > $O/bench_world1_nccl.json 2> $O/bench_world1_nccl.err
GECCO_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline --no-past-l3 > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err
# per-kernel durations of everything outside the C3 step
kt c5 python bench.py --workload C5 --no-past-l3 --no-cpu-baseline --steps 200 --warmup 20 --min-region-ms 0
kt whole_contig python tools/bench_full.py
kt general_l python tools/bench_general.py 3 8 16 32
kt levels python tools/bench_levels.py
# counter traffic of the C5 kernels (separate passes)
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c -d $O/pmc_c5_$n -o pmc -- python tools/bench_full.py > $O/pmc_c5_$n.log 2>&1
done
python tools/prof_summary.py $O "" 2>/dev/null | grep -P "\tPMC\t" | grep -E "f_short|vd_|v_labels|seq_state|seg_|wire_|copy_block" | cut -c1-200 > $O/pmc_c5.txt
rm -rf $O/pmc_c5_*/
tail -1 $O/bench_c3_driver.json | cut -c1-300
ls -la $O | head -50
