# Round-5 latency set, ONE gpurun call:  gpurun --timeout 1500 -- 'bash tools/r5_latency.sh <tag>'
set -u
TAG=${1:-base}
R=$PWD; O=$R/gpurun_out/r5_lat_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 120 tools/ubench/launch_floor > $O/launch_floor.txt 2>&1
timeout 600 python -m benchkit.latency > $O/latency.json 2> $O/latency.err
timeout 120 python -m benchkit.latency --cold-process > $O/cold_process.json 2> $O/cold_process.err
for e in windowed decode clusters; do
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d $O/tl_$e -o tl -- python -m benchkit.latency --loop 12 --loop-entry $e > $O/tl_$e.log 2>&1
  db=$(ls $O/tl_$e/*/*.db $O/tl_$e/*.db 2>/dev/null | head -1)
  python tools/timeline.py $db 2 300 $( [ $e = windowed ] && echo --schema ) > $O/timeline_c1_$e.txt 2>&1
  rm -rf $O/tl_$e
done
tail -c 600 $O/latency.err; cat $O/launch_floor.txt | tail -12; head -c 1500 $O/cold_process.json
