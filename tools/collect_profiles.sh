# copy the summaries of one `tools/r4_profiles.sh` call (gpurun_out/r4_final/) into profiles/ under their committed names
set -eu
O=gpurun_out/r4_final; P=profiles
cp $O/bench_c3_driver.json $P/r04_bench_c3_driver_command.json; cp $O/bench_c3.json $P/r04_bench_c3.json; cp $O/bench_c3_two_launch.json $P/r04_bench_c3_two_launch.json
cp $O/bench_c5.json $P/r04_bench_c5.json; cp $O/bench_c2.json $P/r04_bench_c2.json
cp $O/levels.json $P/r04_levels.json; cp $O/whole_contig.json $P/r04_whole_contig.json; cp $O/general_l.json $P/r04_general_l.json
cp $O/bench_world1_nccl.json $P/r04_bench_world1_nccl.json; cp $O/bench_2ranks_one_device.json $P/r04_bench_2ranks_one_device.json
{ echo "# rocprofv3 --kernel-trace --stats and --pmc passes of bench.py on C3 (tools/profile.sh r4_final); kt = one decode stream, kt2 = two (default schedule)"; cat $O/summary.txt; } > $P/r04_rocprofv3_summary.txt
{ echo "# rocprofv3 --kernel-trace --stats of: bench.py --workload C5 (decode step on 100 x 50 000-gene contigs)"; cat $O/kt_c5.txt; echo; echo "# tools/bench_full.py (rows F and V stand-alone, C3 and C5)"; cat $O/kt_whole_contig.txt; echo; echo "# PMC passes of tools/bench_full.py (FETCH_SIZE / WRITE_SIZE in KiB, averages over the dispatches of both workloads)"; cat $O/pmc_c5.txt; } > $P/r04_c5_rocprofv3_summary.txt
{ echo "# rocprofv3 --kernel-trace --stats of tools/bench_general.py 3 8 16 32 (any-L kernels)"; cat $O/kt_general_l.txt; echo; echo "# ... of tools/bench_levels.py (batch driver: copies, window kernel on chunks, segmenter)"; cat $O/kt_levels.txt; } > $P/r04_general_levels_rocprofv3_summary.txt
python tools/pmc_to_json.py $O/win C3 r04 crf_windowed_l2 > /dev/null
python tools/pmc_to_json.py $O/pipe C3:pipelined r04 crf_decode_pipelined > /dev/null
python tools/pmc_to_json.py $O/c5win C5 r04 crf_windowed_l2 > /dev/null
