# copy the summaries of one `tools/r6_profiles.sh` call (gpurun_out/r6_final/) into profiles/ under their committed names
set -eu
O=gpurun_out/r6_final; P=profiles
line() { grep '^{' "$1" | tail -1; }
line $O/bench_c3_driver.json > $P/r06_bench_c3_driver_command.json; cp $O/bench_c3_driver_detail.json $P/r06_bench_c3_driver_command_detail.json
line $O/bench_c3.json > $P/r06_bench_c3.json; cp $O/bench_c3_detail.json $P/r06_bench_c3_detail.json
for n in c3_two_launch c5 c5_driver c2 c1; do line $O/bench_$n.json > $P/r06_bench_$n.json; cp $O/bench_${n}_detail.json $P/r06_bench_${n}_detail.json; done
line $O/bench_world1_nccl.json > $P/r06_bench_world1_nccl.json; line $O/bench_2ranks_one_device.json > $P/r06_bench_2ranks_one_device.json
for f in levels whole_contig general_l direct_sweep latency cold_process; do line $O/$f.json > $P/r06_$f.json; done
{ echo "# rocprofv3 --kernel-trace --stats and --pmc passes of bench.py on C3 (tools/profile.sh r6_final); kt = one decode stream (begin-to-end durations of launches that run alone: roofline.kernel_us_rocprof), kt2 = two (default schedule: two launches in flight, each longer)"; cat $O/summary.txt; echo "# the pipelined launch with the tiles' score-difference stores off (GECCO_CRF_AB_NO_HANDOVER_STORE=1; labels wrong): EXPERIMENTS.md, round 6, item 6"; sed 's/^/nostore\//' $O/nostore_summary.txt; } > $P/r06_rocprofv3_summary.txt
{ echo "# rocprofv3 --kernel-trace --stats of: bench.py --workload C5 (decode step on 100 x 50 000-gene contigs; long-contig Viterbi = vd_fold, vd_replay, v_labels_refine, vd_exact_fix)"; cat $O/kt_c5.txt; echo; echo "# tools/bench_full.py (rows F and V stand-alone, C3 and C5)"; cat $O/kt_whole_contig.txt; } > $P/r06_c5_rocprofv3_summary.txt
{ echo "# rocprofv3 --kernel-trace --stats of tools/bench_general.py 3 8 16 32 (any-L kernels)"; cat $O/kt_general_l.txt; echo; echo "# ... of tools/bench_levels.py (batch driver: copies, window kernel on chunks, segmenter)"; cat $O/kt_levels.txt; } > $P/r06_general_levels_rocprofv3_summary.txt
{ echo "# tools/ubench/launch_floor (back-to-back launches; launch + wait latency: the floor of a synchronous one-shot call)"; cat $O/launch_floor.txt; echo; echo "# tools/ubench/pcie_bw"; cat $O/pcie_bw.txt; } > $P/r06_launch_floor_pcie.txt
{ echo "# one WARM C1 call through the batch driver's direct path, per entry point (rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace of python -m benchkit.latency --loop, tools/timeline.py)"; for e in windowed decode clusters; do echo "== $e"; cat $O/timeline_c1_$e.txt; echo; done; } > $P/r06_c1_timeline.txt
cp $O/reference_bits.jsonl $P/r06_reference_bits.jsonl; cp $O/kt_reference_bits.txt $P/r06_reference_bits_rocprofv3_summary.txt
{ echo "# tools/host_issue_probe.py: what ONE launch costs the host (enqueue loop, no wait inside), batches the GPU is never the limit of"; grep -v amdgpu $O/host_issue.txt; echo; echo "# tools/r6_streams.sh: decode streams in flight, step per batch (us)"; cat $O/streams.txt; echo; echo "# tools/r6_handover_ab.sh: the step with the WRITE half of the hand-over between launches off (wrong labels)"; grep -v amdgpu $O/handover_ab.txt; echo; echo "# tests/test_gpu_reference_bits.py -k c3"; cat $O/reference_bits_vs_libm.txt; } > $P/r06_ab_raw.txt
python tools/pmc_to_json.py $O/win C3 r06 crf_windowed_l2 > /dev/null
python tools/pmc_to_json.py $O/pipe C3:pipelined r06 crf_decode_pipelined > /dev/null
python tools/pmc_to_json.py $O/c5win C5 r06 crf_windowed_l2 > /dev/null
python tools/pmc_to_json.py $O/nostore C3:pipelined:no_handover_store r06 crf_decode_pipelined > /dev/null
python tools/kt_to_json.py $O/kt C3:pipelined r06 crf_decode_pipelined
python tools/kt_to_json.py $O/ktw C3 r06 crf_windowed_l2 "rocprofv3 --kernel-trace --stats of bench.py --streams 1 --windowed-only (plain windowed launches)"
python -c "import json; d=json.load(open('$P/pmc_traffic.json')); print({k: (v.get('kernel_source_sha16'), v.get('hbm_bytes_per_launch'), v.get('kernel_us_rocprof')) for k, v in d.items() if isinstance(v, dict)})"
