#!/usr/bin/env python3
"""Same-box A/B of the batch driver's levels on C3 under environment switches:
    python tools/levels_ab.py VAR=a VAR=b [...]     (each setting runs in its own process, twice, interleaved)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
import torch
from gecco_amd import _native as nat, synth
from benchkit import levels
wl = synth.workload("C3")
model = nat.Model.from_tables(wl["w"], wl["trans"])
lv = levels.host_buffer_levels(model, wl, devices=(0,))
print("AB", json.dumps({k: round(v["ms"], 4) for k, v in lv.items() if isinstance(v, dict) and "ms" in v}))
""" % ROOT

if __name__ == "__main__":
    settings = sys.argv[1:] or ["X=0"]
    for rep in range(2):
        for st in settings:
            k, v = st.split("=", 1)
            out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **{k: v}), capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("AB ")]
            print(st, line[-1][3:] if line else out.stderr[-400:])
