#!/usr/bin/env python3
"""PCIe-inclusive rate of the ndarray (host) path: evaluate() on NumPy arrays, staged by
fcamd_evaluate_host.  Reported in DESIGN.md next to the device-resident number; never the
bench `value`.  Registered (page-locked) arrays take the zero-copy path; FCAMD_ZERO_COPY=0 in the
environment keeps them on the staged path (the A/B of DESIGN.md)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}


def run(n, register):
    rng = np.random.default_rng(0)
    law = fc.VonMises3D(VM_P)
    g = rng.normal(size=9 * n) * np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)
    s0 = np.zeros(6 * n)
    t = np.zeros(36 * n)
    h0 = {"eps_n": np.zeros(6 * n), "alpha": rng.uniform(0, 0.02, size=n)}
    s, h = s0.copy(), {k: v.copy() for k, v in h0.items()}
    arrays = [g, s, t, h["eps_n"], h["alpha"]]
    ctx = law._handle(0).ctx
    if register:
        for a in arrays:
            ctx.register_host_buffer(a)
    times = []
    for _ in range(4):
        s[:] = s0
        for k in h:
            h[k][:] = h0[k]
        t0 = time.perf_counter()
        law.evaluate(0.0, 1.0, g, s, t, h)
        times.append(time.perf_counter() - t0)
    mode = ctx.last_host_mode()
    if register:
        for a in arrays:
            ctx.unregister_host_buffer(a)
    best = min(times[1:])
    return {"n": n, "registered": register, "zero_copy_mode": mode, "ms": round(best * 1e3, 2), "Mpts_s": round(n / best / 1e6, 1),
            "GB_s_pcie": round(n * 568 / best / 1e9, 2)}


def run_resident(n, register):
    """ResidentState.evaluate_into: grad up, stress + tangent down, state on the device."""
    from fenics_constitutive_amd.resident import ResidentState

    rng = np.random.default_rng(0)
    law = fc.VonMises3D(VM_P)
    g = rng.normal(size=9 * n) * np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)
    s, t = np.zeros(6 * n), np.zeros(36 * n)
    st = ResidentState(law, n, history0={"eps_n": np.zeros(6 * n), "alpha": rng.uniform(0, 0.02, size=n)})
    ctx = law._handle(0).ctx
    if register:
        for a in (g, s, t):
            ctx.register_host_buffer(a)
    times = []
    for _ in range(4):
        t0 = time.perf_counter()
        st.evaluate_into(0.0, 1.0, g, s, t)
        times.append(time.perf_counter() - t0)
    mode = ctx.last_host_mode()
    if register:
        for a in (g, s, t):
            ctx.unregister_host_buffer(a)
    best = min(times[1:])
    return {"path": "resident", "n": n, "registered": register, "zero_copy_mode": mode, "ms": round(best * 1e3, 2),
            "Mpts_s": round(n / best / 1e6, 1), "GB_s_pcie": round(n * 408 / best / 1e9, 2)}


if __name__ == "__main__":
    ns = (10_000_000,) if "--big" in sys.argv else (1_000_000, 10_000_000)
    for n in ns:
        for reg in (False, True):
            print(json.dumps(run(n, reg)), flush=True)
            print(json.dumps(run_resident(n, reg)), flush=True)
