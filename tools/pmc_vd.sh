# usage (GPU box): source tools/pmc_vd.sh; pmc_vd <tag>   -> VALU / SALU / LDS instruction counts of vd_short (two-launch schedule)
pmc_vd() {
  local tag=$1; shift
  local O=$PWD/gpurun_out/pmcvd_$tag; mkdir -p $O
  ( cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES -d $O -o pmc -- python $OLDPWD/bench.py --no-cpu-baseline --no-past-l3 --no-c4 --no-8d --no-levels --schedule two-launch --steps 3 --warmup 1 --kernel-iters 3 --preroll-ms 0 > $O/log.txt 2>&1 )
  python tools/prof_summary.py $O vd_short | sed "s/^/$tag /" | cut -c1-120
}
