#!/bin/bash
# Run on the GPU box: the PMC passes that say where a window kernel's cycles go, for one build / switch setting.
# usage: [ENV=...] tools/pmc_ab.sh <tag>      -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
B="python bench.py --no-cpu-baseline --no-past-l3 --steps 3 --warmup 1 --kernel-iters 3 --preroll-ms 0 --windowed-only --min-region-ms 0 $*"
timeout 180 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc1 -o pmc1 -- $B > $O/pmc1.log 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM -d $O/pmc4 -o pmc4 -- $B > $O/pmc4.log 2>&1
timeout 180 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU -d $O/pmc7 -o pmc7 -- $B > $O/pmc7.log 2>&1
python tools/prof_summary.py $O crf_ > $O/summary.txt 2>&1
cat $O/summary.txt
