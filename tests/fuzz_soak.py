#!/usr/bin/env python3
"""Soak run of the randomised protocol tests with seeds beyond the three the suite runs:
   python tests/fuzz_soak.py [first seed] [last seed]      (one process; prints one line per failure)"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")  # as tests/conftest.py
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]  # lives under tests/: the test modules it runs import the oracle
import test_gpu_multidevice_fuzz as MF  # noqa: E402
import test_gpu_problem_fuzz as PF  # noqa: E402
import test_gpu_resident_fuzz as RF  # noqa: E402

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3, 40)
kinds = ["von_mises_3d", "von_mises_3d+unpacked", "von_mises_3d+dense_rows", "comfe_mises_plasticity", "comfe_mises_plasticity+unpacked",
         "comfe_mises_plasticity+rows7", "drucker_prager", "drucker_prager+unpacked", "drucker_prager_hyperbolic",
         "drucker_prager_hyperbolic+unpacked", "linear_elasticity", "spring_maxwell"]
bad = 0
for seed in range(lo, hi + 1):
    cases = [(RF.test_random_call_sequences, (k, seed)) for k in kinds] + [(PF.test_random_call_sequences, (seed, True)), (PF.test_random_call_sequences, (seed, False))]
    cases += [(MF.test_random_call_sequences, (k, [0] * (2 + (seed + i) % 3), seed)) for i, k in enumerate(
        ["von_mises_3d", "comfe_mises_plasticity", "comfe_mises_plasticity+rows7", "drucker_prager", "drucker_prager_hyperbolic",
         "linear_elasticity", "spring_maxwell", "spring_kelvin"])]
    for fn, args in cases:
        try:
            fn(*args)
        except Exception:  # noqa: BLE001
            bad += 1
            print("FAIL", fn.__module__, args, traceback.format_exc().splitlines()[-1], flush=True)
    print("seed", seed, "done", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
