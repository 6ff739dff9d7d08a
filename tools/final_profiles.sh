# Round-3 profile set, ONE gpurun call:  /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/final_profiles.sh'
set -u
R=$PWD; O=$R/gpurun_out/r3_final; mkdir -p $O
timeout 900 tools/profile.sh r3_final > $O/profile.log 2>&1
timeout 400 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
timeout 300 python bench.py --schedule two-launch --no-levels --no-cpu-baseline > $O/bench_c3_two_launch.json 2> $O/bench_c3_two_launch.err
timeout 300 python bench.py --workload C5 --no-past-l3 > $O/bench_c5.json 2> $O/bench_c5.err
GECCO_CRF_VD_EXACT=0 timeout 300 python bench.py --workload C5 --no-past-l3 --no-cpu-baseline > $O/bench_c5_vd_exact0.json 2> $O/bench_c5_vd_exact0.err
timeout 400 python tools/bench_levels.py > $O/levels.json 2> $O/levels.err
timeout 300 python tools/bench_full.py > $O/full.json 2> $O/full.err
timeout 400 python tools/bench_general.py > $O/general.json 2> $O/general.err
GECCO_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline --no-past-l3 > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err
  for lag in 176 100000; do
      timeout 200 python bench.py --schedule two-launch --no-cpu-baseline --no-levels --no-8d --no-past-l3 --no-c4 --steps 20 --warmup 5 --kernel-iters 2 > /dev/null 2>&1
  done
fi
tail -1 $O/bench_c3.json | cut -c1-400
tail -3 $O/profile.log
ls -la $O | head -40
