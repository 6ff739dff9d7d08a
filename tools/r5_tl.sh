set -u
R=$PWD; O=$R/gpurun_out/r5_tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for mode in torch notorch; do
  E=""; [ $mode = notorch ] && E="NO_TORCH=1"
  env $E GECCO_CRF_P_TO_HOST=0 timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl_$mode -o tl -- python tools/trace_decode.py > $O/tl_$mode.log 2>&1
  db=$(ls $O/tl_$mode/*/*.db $O/tl_$mode/*.db 2>/dev/null | head -1)
  python tools/timeline.py $db 2 1000 > $O/timeline_decode_$mode.txt 2>&1
  grep "^call" $O/tl_$mode.log | tail -3
  rm -rf $O/tl_$mode
done
for mode in torch notorch; do echo "== $mode"; cat $O/timeline_decode_$mode.txt | tail -45; done
