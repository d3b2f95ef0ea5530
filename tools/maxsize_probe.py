#!/usr/bin/env python3
"""Largest single-GPU problem: LinearElasticityModel at n = 4.2e8 points (tangent = 1.5e10 doubles,
121 GB; byte offsets beyond 2^36).  Checks size-independent properties: the tangent of every point is
D, the stress increment is linear in the gradient, and the last tile is written."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 420_000_037
dev = torch.device("cuda", 0)
law = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL)
g = torch.empty(9 * n, dtype=torch.float64, device=dev).normal_(generator=torch.Generator(device=dev).manual_seed(1)).mul_(1e-3)
s1 = torch.zeros(6 * n, dtype=torch.float64, device=dev)
t = torch.full((36 * n,), float("nan"), dtype=torch.float64, device=dev)
law.evaluate(0.0, 1.0, g, s1, t, None)
torch.cuda.synchronize()
D = torch.from_numpy(np.asarray(law.D).reshape(-1)).to(dev)
tv = t.view(n, 36)
bad = 0
for a in range(0, n, 50_000_000):  # chunked: no 121 GB temporaries
    bad += int((tv[a:a + 50_000_000] != D).any(dim=1).sum())
print("points whose tangent != D:", bad)
# linearity: evaluating with the same gradient again doubles the stress (sigma += D eps)
s2 = s1.clone()
law.evaluate(0.0, 1.0, g, s2, None, None)
err = 0.0
for a in range(0, 6 * n, 300_000_000):
    err = max(err, float((s2[a:a + 300_000_000] - 2 * s1[a:a + 300_000_000]).abs().max()))
print("max |sigma(2 steps) - 2 sigma(1 step)|:", err, " max |sigma|:", float(s1.abs().max()))
# the last points against the formula
k = 5
ge = g.view(n, 9)[-k:].cpu().numpy()
eps = np.stack([ge[:, 0], ge[:, 4], ge[:, 8], (ge[:, 1] + ge[:, 3]) / 2**0.5, (ge[:, 2] + ge[:, 6]) / 2**0.5, (ge[:, 5] + ge[:, 7]) / 2**0.5], axis=1)
ref = eps @ np.asarray(law.D)
print("tail rel err:", float(np.abs(s1.view(n, 6)[-k:].cpu().numpy() - ref).max() / np.abs(ref).max()))
assert bad == 0 and err <= 1e-12 * float(s1.abs().max()) + 1e-18
print("ok n =", n)
