#!/usr/bin/env python3
"""Uniaxial tension of n material points with VonMises3D, state resident on the GPU.

The loop is the one `IncrSmallStrainProblem` drives (tests/models/test_plasticity.py:13-137 of the
reference, one unit cube): per load increment, Newton iterations on the lateral strains until the
lateral stresses vanish -- every iteration is one `ResidentState.evaluate_into` (gradient up, trial
stress + consistent tangent down, committed state untouched on the device), every converged increment
one pointer-swap `update()`.

    python examples/uniaxial_tension_resident.py [n_points]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
state = ResidentState(law, n)
max_strain = 0.05 * np.linspace(0.6, 1.0, n)  # every point its own load amplitude
stress, tangent = np.zeros(6 * n), np.zeros(36 * n)
grad = np.zeros((n, 9))
prev = np.zeros(n)
iterations = 0
for step in np.linspace(0, 1, 101)[1:]:
    cur = step * max_strain
    d_eps = np.zeros((n, 3))
    d_eps[:, 0] = cur - prev
    while True:
        grad[:, 0], grad[:, 4], grad[:, 8] = d_eps[:, 0], d_eps[:, 1], d_eps[:, 2]
        state.evaluate_into(0.0, 1.0, grad.reshape(-1), stress, tangent)
        s = stress.reshape(n, 6)
        r = s[:, 1:3]
        if np.abs(r).max() < 1e-9:
            break
        J = tangent.reshape(n, 6, 6)[:, 1:3, 1:3]
        d_eps[:, 1:3] -= np.linalg.solve(J, r[:, :, None])[:, :, 0]
        iterations += 1
    state.update()
    prev = cur
    if round(step * 100) % 20 == 0:
        print(f"load {step:4.2f}: sigma_xx in [{s[:, 0].min():8.2f}, {s[:, 0].max():8.2f}], "
              f"plastic points {int(law.last_stats.n_plastic)}/{n}")
print(f"{n} points, 100 increments, {iterations} Newton iterations; max sigma_xx = {s[:, 0].max():.3f} (y_inf = 2500)")
assert s[:, 0].max() <= 2500.0 + 1e-8
