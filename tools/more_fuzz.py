#!/usr/bin/env python3
"""The randomised GPU sweeps of tests/ on seeds beyond the committed ones (run on the GPU box):
    python tools/more_fuzz.py [first_seed] [count]     (batch driver: tests/test_gpu_fuzz_session.py; objects: test_gpu_fuzz_objects.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from gecco_amd import _native as nat  # noqa: E402
import tests.test_gpu_fuzz_objects as fo  # noqa: E402
import tests.test_gpu_fuzz_session as fs  # noqa: E402
from oracle import lcrf  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 300
count = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
om = lcrf.load_model(os.path.join(ROOT, "tests", "golden", "model.pkl"), os.path.join(ROOT, "tests", "golden", "model.pkl.md5"))
bad = 0
for seed in range(first, first + count):
    for name, fn in (("session", lambda: fs.test_session_against_the_oracle_on_a_random_case(nat, seed)),
                     ("objects", (lambda: fo.test_predict_probabilities_on_random_objects(om, seed)) if seed % 10 == 0 else None)):
        if fn is None:
            continue
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("SEED", name, seed, type(e).__name__, str(e)[:300])
    if bad > 5:
        break
print(f"seeds {first} .. {first + count - 1}: failures {bad}")
