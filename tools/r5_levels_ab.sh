set -u
R=$PWD; O=$R/gpurun_out/r5_ab; mkdir -p $O
python -m pytest tests/test_gpu_session.py tests/test_gpu_direct.py tests/test_gpu_multidevice.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -2
for r in 1 2; do
for v in "GECCO_CRF_Y_TO_HOST=1" "GECCO_CRF_Y_TO_HOST=0" "GECCO_CRF_TAPER=1" "GECCO_CRF_CHUNK_GENES=262144" "GECCO_CRF_CHUNK_GENES=393216"; do
env $v python - <<PY
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from gecco_amd import _native as nat, synth, levels
wl = synth.workload("C3")
model = nat.Model.from_tables(wl["w"], wl["trans"])
lv = levels.host_buffer_levels(model, wl, devices=(0,), reps=9)
print("AB $v", {k: round(v["ms"], 3) for k, v in lv.items() if k.startswith(("one_shot_pinned", "decode"))})
PY
done; done 2>&1 | grep "^AB" | tee $O/levels_ab.txt
