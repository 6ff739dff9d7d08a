# usage (on the GPU box): source tools/kt.sh; kt <tag> [bench args]  -> per-kernel average durations of one bench.py run
kt() {
  local tag=$1; shift
  local O=$PWD/gpurun_out/kt_$tag; mkdir -p $O
  ( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats -d $O -o kt -- python $OLDPWD/bench.py --no-cpu-baseline --no-past-l3 --no-c4 --no-8d --no-levels --streams 1 --steps 500 --warmup 50 --kernel-iters 50 "$@" > $O/log.txt 2>&1 )
  python tools/prof_summary.py $O "" 2>/dev/null | grep KERNEL | sed "s/^/$tag /" | cut -c1-220
}
