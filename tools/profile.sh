#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + separate PMC passes of bench.py.
# usage: tools/profile.sh <tag> [extra bench args]
# Outputs under gpurun_out/<tag>/ ; summary in gpurun_out/<tag>/summary.txt
set -u
TAG=${1:-prof}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
B="python bench.py --no-cpu-baseline --no-past-l3 --no-c4 --no-8d --no-levels --streams 1 $*"
# kernel trace of the bench run on ONE decode stream (device pre-roll + 100 warmup + 1000 timed steps + 200 kernel timings):
# per-kernel averages that are launch durations
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B > $O/kt.log 2>&1
# the same run on the default schedule (two decode streams: two launches of crf_decode_pipelined in flight, each longer than alone)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt2 -o kt2 -- ${B/--streams 1/--streams 2} > $O/kt2.log 2>&1
# ... and with plain windowed launches only (`--windowed-only`): in the two runs above the window kernel's symbol also serves the
# two-launch schedule's first launch, which writes the score differences too (~31 us): its average there mixes the two (29.3 us
# in round 5 and 6 against 26 us by events -- the gap the round-5 review asked about)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktw -o ktw -- $B --windowed-only > $O/ktw.log 2>&1
S="--steps 3 --warmup 1 --kernel-iters 3 --preroll-ms 0 --windowed-only --no-past-l3 --min-region-ms 0"  # few dispatches of ONE kind (plain windowed launches): the PMC passes serialise and slow every launch
timeout 180 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc1 -o pmc1 -- $B $S > $O/pmc1.log 2>&1
timeout 180 rocprofv3 --pmc FETCH_SIZE -d $O/pmc2 -o pmc2 -- $B $S > $O/pmc2.log 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE -d $O/pmc3 -o pmc3 -- $B $S > $O/pmc3.log 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM -d $O/pmc4 -o pmc4 -- $B $S > $O/pmc4.log 2>&1
timeout 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/pmc5 -o pmc5 -- $B $S > $O/pmc5.log 2>&1
timeout 180 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/pmc6 -o pmc6 -- $B $S > $O/pmc6.log 2>&1
# the same for the launch that is the step under the default (pipelined) schedule: crf_decode_pipelined
S2="--steps 4 --warmup 1 --kernel-iters 3 --preroll-ms 0 --min-region-ms 0"
timeout 180 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pipe/pmc1 -o pmc1 -- $B $S2 > $O/pipe_pmc1.log 2>&1
timeout 180 rocprofv3 --pmc FETCH_SIZE -d $O/pipe/pmc2 -o pmc2 -- $B $S2 > $O/pipe_pmc2.log 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE -d $O/pipe/pmc3 -o pmc3 -- $B $S2 > $O/pipe_pmc3.log 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM -d $O/pipe/pmc4 -o pmc4 -- $B $S2 > $O/pipe_pmc4.log 2>&1
python tools/prof_summary.py $O crf_ > $O/summary.txt 2>&1
mkdir -p $O/win; for k in 1 2 3 4 5 6; do mv $O/pmc$k $O/win/ 2>/dev/null; done
python tools/pmc_to_json.py $O/win C3 $TAG crf_windowed_l2 > $O/pmc.json 2>&1
python tools/pmc_to_json.py $O/pipe C3:pipelined $TAG crf_decode_pipelined >> $O/pmc.json 2>&1
# the committed begin-to-end durations bench.py quotes as roofline.kernel_us_rocprof (one decode stream: a launch alone on the chip)
python tools/kt_to_json.py $O/kt C3:pipelined $TAG crf_decode_pipelined >> $O/pmc.json 2>&1
python tools/kt_to_json.py $O/ktw C3 $TAG crf_windowed_l2 "rocprofv3 --kernel-trace --stats of bench.py --streams 1 --windowed-only (plain windowed launches)" >> $O/pmc.json 2>&1
cat $O/summary.txt
