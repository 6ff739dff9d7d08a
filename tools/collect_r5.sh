# copy the summaries of one `tools/r5_profiles.sh` call (gpurun_out/r5_final/) into profiles/ under their committed names
set -eu
O=gpurun_out/r5_final; P=profiles
line() { grep '^{' "$1" | tail -1; }
line $O/bench_c3_driver.json > $P/r05_bench_c3_driver_command.json; line $O/bench_c3.json > $P/r05_bench_c3.json
line $O/bench_c3_two_launch.json > $P/r05_bench_c3_two_launch.json
line $O/bench_c5.json > $P/r05_bench_c5.json; line $O/bench_c5_driver.json > $P/r05_bench_c5_driver_command.json
line $O/bench_c2.json > $P/r05_bench_c2.json; line $O/bench_c1.json > $P/r05_bench_c1.json
line $O/bench_world1_nccl.json > $P/r05_bench_world1_nccl.json; line $O/bench_2ranks_one_device.json > $P/r05_bench_2ranks_one_device.json
for f in levels whole_contig general_l direct_sweep latency cold_process; do line $O/$f.json > $P/r05_$f.json; done
{ echo "# rocprofv3 --kernel-trace --stats and --pmc passes of bench.py on C3 (tools/profile.sh r5_final); kt = one decode stream, kt2 = two (default schedule)"; cat $O/summary.txt; } > $P/r05_rocprofv3_summary.txt
{ echo "# rocprofv3 --kernel-trace --stats of: bench.py --workload C5 (decode step on 100 x 50 000-gene contigs; long-contig Viterbi = vd_fold, vd_replay, v_labels_refine, vd_exact_fix)"; cat $O/kt_c5.txt; echo; echo "# tools/bench_full.py (rows F and V stand-alone, C3 and C5)"; cat $O/kt_whole_contig.txt; } > $P/r05_c5_rocprofv3_summary.txt
{ echo "# rocprofv3 --kernel-trace --stats of tools/bench_general.py 3 8 16 32 (any-L kernels; gl_chunk_rows_mfma = round 5)"; cat $O/kt_general_l.txt; echo; echo "# ... of tools/bench_levels.py (batch driver: copies, window kernel on chunks, segmenter)"; cat $O/kt_levels.txt; } > $P/r05_general_levels_rocprofv3_summary.txt
{ echo "# tools/ubench/launch_floor (back-to-back launches; launch + wait latency: the floor of a synchronous one-shot call)"; cat $O/launch_floor.txt; echo; echo "# tools/ubench/pcie_bw (C3's wire: 19.3 MB up, 16 MB down; alone, at once, in four chunks; copy engine against a kernel's stores)"; cat $O/pcie_bw.txt; } > $P/r05_launch_floor_pcie.txt
{ echo "# one WARM C1 call (one 50-gene contig, pretrained weights) through the batch driver's direct path, per entry point -- rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace of python -m benchkit.latency --loop, tools/timeline.py (A = HIP API call on the host, K = kernel, C = copy)"; for e in windowed decode clusters; do echo "== $e"; cat $O/timeline_c1_$e.txt; echo; done; } > $P/r05_c1_timeline.txt
{ echo "# one C3 decode call (2 M genes, pinned buffers) through the chunked batch driver: kernels and copies, PyTorch's HIP runtime / the system's"; echo "== torch"; cat $O/timeline_decode_torch.txt; echo; echo "== system runtime"; cat $O/timeline_decode_notorch.txt; } > $P/r05_c3_decode_timeline.txt
cp $O/multi_entry.txt $P/r05_multi_entry.txt
cp $O/reference_bits.jsonl $P/r05_reference_bits.jsonl; cp $O/kt_reference_bits.txt $P/r05_reference_bits_rocprofv3_summary.txt
cp $O/tiles_sweep.txt $P/r05_tiles_sweep.txt
python tools/pmc_to_json.py $O/win C3 r05 crf_windowed_l2 > /dev/null
python tools/pmc_to_json.py $O/pipe C3:pipelined r05 crf_decode_pipelined > /dev/null
python tools/pmc_to_json.py $O/c5win C5 r05 crf_windowed_l2 > /dev/null
python -c "import json; d=json.load(open('$P/pmc_traffic.json')); print({k: (v.get('kernel_source_sha16'), v.get('hbm_bytes_per_launch')) for k, v in d.items() if isinstance(v, dict)})"
