set -u
R=$PWD; O=$R/gpurun_out/r2_final3; mkdir -p $O
timeout 600 tools/profile.sh r2_final3 > $O/profile.log 2>&1
timeout 300 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
timeout 300 python bench.py --workload C5 --no-past-l3 > $O/bench_c5.json 2> $O/bench_c5.err
timeout 400 python tools/bench_levels.py > $O/levels.json 2> $O/levels.err
timeout 300 python tools/bench_full.py > $O/full.json 2> $O/full.err
timeout 400 python tools/bench_general.py > $O/general.json 2> $O/general.err
GECCO_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline --no-past-l3 > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err
tail -1 $O/bench_c3.json | cut -c1-600
tail -2 $O/profile.log
ls -la $O | head -30
