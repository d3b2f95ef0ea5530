#!/usr/bin/env python3
"""What the separate tail kernel (n % 64 points) costs a small ndarray call: median us per `evaluate` at n = 1024 (one launch) and
n = 1000 (main kernel + tail kernel), LinearElasticity and VonMises3D."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FCAMD_SMALL_CALL_WARNING"] = "0"
import fenics_constitutive_amd as fc  # noqa: E402

FULL = fc.StressStrainConstraint.FULL
laws = {"LinearElasticityModel": (fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, FULL), None),
        "VonMises3D": (fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}), ("eps_n", "alpha"))}
rng = np.random.default_rng(0)
for name, (law, hk) in laws.items():
    for n in (960, 1000, 1024, 4096, 4100):
        g, s, t = rng.normal(scale=1e-3, size=9 * n), np.zeros(6 * n), np.zeros(36 * n)
        h = None if hk is None else {"eps_n": np.zeros(6 * n), "alpha": np.zeros(n)}
        for _ in range(20):
            law.evaluate(0.0, 1.0, g, s, t, h)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            law.evaluate(0.0, 1.0, g, s, t, h)
            ts.append(time.perf_counter() - t0)
        print(f"{name} n={n} (n % 64 = {n % 64}): median {sorted(ts)[100] * 1e6:.1f} us", flush=True)
