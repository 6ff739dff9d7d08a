set -u
R=$PWD; O=$R/gpurun_out/r5_refbits; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout 900 python tools/refbits_bench.py > $O/refbits.log 2>&1; echo "refbits rc=$?"; cat $O/refbits.log | cut -c1-600
GECCO_CRF_REFERENCE_BITS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_ref -o kt -- python tools/refbits_bench.py resident > $O/kt_ref.log 2>&1
python tools/prof_summary.py $O/kt_ref 2>/dev/null | head -20
ls $O/kt_ref/* | head
