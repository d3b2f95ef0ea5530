#!/usr/bin/env python3
"""The uniaxial-tension loop of examples/uniaxial_tension_resident.py with ONE process driving several GPUs.

The assembler's arrays (gradient, stress, tangent) are NumPy arrays of this process; every GPU keeps the committed and
trial state of its own contiguous slice of the points in its HBM and, per Newton iteration, reads its slice of the
gradient from and writes its slice of stress / tangent to those arrays over its own PCIe link
(`MultiDeviceResidentState` -> `fcamd_multi_state_evaluate`; nothing is gathered, nothing crosses xGMI).  Results are
bit-identical to one GPU.  On a box with fewer GPUs than asked for, several contexts share a device.

    python examples/uniaxial_tension_multi_gpu.py [n_points] [n_devices]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device count only)

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.multidevice import MultiDeviceResidentState  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
want = int(sys.argv[2]) if len(sys.argv) > 2 else max(2, torch.cuda.device_count())
devices = [k % torch.cuda.device_count() for k in range(want)]
law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
state = MultiDeviceResidentState(law, n, devices=devices)
print("devices", devices, "slices", state.slices())
max_strain = 0.05 * np.linspace(0.6, 1.0, n)
stress, tangent = np.zeros(6 * n), np.zeros(36 * n)
grad = np.zeros((n, 9))
state.pin_host_arrays(grad.reshape(-1), stress, tangent)  # the same three arrays every iteration: page-lock them once
prev = np.zeros(n)
iterations = 0
for step in np.linspace(0, 1, 51)[1:]:
    cur = step * max_strain
    d_eps = np.zeros((n, 3))
    d_eps[:, 0] = cur - prev
    while True:
        grad[:, 0], grad[:, 4], grad[:, 8] = d_eps[:, 0], d_eps[:, 1], d_eps[:, 2]
        st = state.evaluate_into(0.0, 1.0, grad.reshape(-1), stress, tangent)
        s = stress.reshape(n, 6)
        r = s[:, 1:3]
        if np.abs(r).max() < 1e-9:
            break
        J = tangent.reshape(n, 6, 6)[:, 1:3, 1:3]
        d_eps[:, 1:3] -= np.linalg.solve(J, r[:, :, None])[:, :, 0]
        iterations += 1
    state.update()
    prev = cur
    if round(step * 50) % 10 == 0:
        print(f"load {step:4.2f}: sigma_xx in [{s[:, 0].min():8.2f}, {s[:, 0].max():8.2f}], plastic points {int(st.n_plastic)}/{n}")
print(f"{n} points on {len(devices)} device contexts, 50 increments, {iterations} Newton iterations; "
      f"max sigma_xx = {s[:, 0].max():.3f} (y_inf = 2500)")
assert s[:, 0].max() <= 2500.0 + 1e-8
state.close()
