#!/usr/bin/env python3
"""Per-call time of evaluate() across problem sizes, device path and host (ndarray) path."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
law = fc.VonMises3D(VM_P)
le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL)
rng = np.random.default_rng(0)
for n in (1_000, 10_000, 100_000, 1_000_000, 10_000_000):
    g = rng.normal(size=9 * n) * np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)
    s, t = np.zeros(6 * n), np.zeros(36 * n)
    h = {"eps_n": np.zeros(6 * n), "alpha": rng.uniform(0, 0.02, size=n)}
    gd, sd, td = torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
    hd = {k: torch.from_numpy(v).cuda() for k, v in h.items()}
    reps = 200 if n <= 100_000 else 20
    out = {"n": n}
    s1 = torch.empty_like(sd)
    h1 = {k: torch.empty_like(v) for k, v in hd.items()}
    law.evaluate_from(0, 1, gd, sd, s1, td, hd, h1)
    st = law.device_stats()
    out["plastic_frac"] = round(st.n_plastic / n, 3)
    out["newton_its"] = round(st.n_newton_iters / max(st.n_plastic, 1), 2)
    for name, fn in (("vm_device", lambda: law.evaluate_from(0, 1, gd, sd, s1, td, hd, h1)), ("le_device", lambda: le.evaluate(0, 1, gd, sd, td, None)),
                     ):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name + "_us"] = round(dt * 1e6, 1)
        out[name + "_Mpts_s"] = round(n / dt / 1e6, 1)
    print(json.dumps(out), flush=True)
