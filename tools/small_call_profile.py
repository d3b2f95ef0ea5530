#!/usr/bin/env python3
"""Where one small ndarray evaluate spends its time: wall per call, the C entry alone (ctypes), cProfile of the Python side.
    python tools/small_call_profile.py [n]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FCAMD_SMALL_CALL_WARNING", "0")
import fenics_constitutive_amd as fc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(0)
for name, law, hd in (("LinearElasticityModel", fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL), None),
                      ("VonMises3D", fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}), {"eps_n": 6, "alpha": 1})):
    g, s, t = rng.normal(scale=1e-3, size=9 * n), np.zeros(6 * n), np.zeros(36 * n)
    h = None if hd is None else {k: np.zeros(d * n) for k, d in hd.items()}
    for _ in range(20):
        law.evaluate(0.0, 1.0, g, s, t, h)
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        law.evaluate(0.0, 1.0, g, s, t, h)
    wall = (time.perf_counter() - t0) / reps * 1e6
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        law.evaluate(0.0, 1.0, g, s, t, h)
    pr.disable()
    st = pstats.Stats(pr)
    total = sum(v[2] for v in st.stats.values())  # tottime
    rows = sorted(((v[2] / reps * 1e6, v[0] // reps, f"{os.path.basename(k[0])}:{k[1]} {k[2]}") for k, v in st.stats.items()), reverse=True)[:12]
    print(f"== {name}, n = {n}: {wall:.1f} us per call (under cProfile: {total / reps * 1e6:.1f} us of own time)")
    for us, calls, where in rows:
        print(f"   {us:7.2f} us  x{calls:<3d} {where}")
