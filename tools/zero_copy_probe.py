#!/usr/bin/env python3
"""Experiment: the evaluate kernel launched DIRECTLY on page-locked host arrays (the GPU reads the
inputs and writes the results over PCIe itself) against the staged host path (fcamd_evaluate_host:
chunked H2D -> kernel -> D2H).  Same law, same inputs, results compared bit for bit.

  (a) staged, registered NumPy arrays                 -- what ships
  (b) zero copy, torch pinned tensors (hipHostMalloc)  -- host pointer == device pointer
  (c) zero copy, hipHostRegister'ed NumPy arrays       -- device pointer from hipHostGetDevicePointer
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd import _capi  # noqa: E402

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}


def hip_runtime():
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return C.CDLL(line.split()[-1])
    raise RuntimeError("HIP runtime not mapped")


def inputs(n):
    rng = np.random.default_rng(0)
    g = rng.normal(size=9 * n) * np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)
    return g, rng.normal(size=6 * n) * 100.0, rng.uniform(0, 0.02, size=n)


def staged(law, n, g, s0, a0, reps):
    s, t = s0.copy(), np.zeros(36 * n)
    h = {"eps_n": np.zeros(6 * n), "alpha": a0.copy()}
    arrays = [g, s, t, h["eps_n"], h["alpha"]]
    ctx = law._handle(0).ctx
    for a in arrays:
        ctx.register_host_buffer(a)
    best = None
    for _ in range(reps):
        s[:] = s0
        h["eps_n"][:] = 0.0
        h["alpha"][:] = a0
        t0 = time.perf_counter()
        law.evaluate(0.0, 1.0, g, s, t, h)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    for a in arrays:
        ctx.unregister_host_buffer(a)
    return best, (s, t, h["eps_n"], h["alpha"])


def zero_copy_run(law, n, ptrs, reset, reps):
    m = law._handle(0)
    m.ctx.set_stream(torch.cuda.current_stream(0).cuda_stream)
    best = None
    for _ in range(reps):
        reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.evaluate_device(0.0, 1.0, n, ptrs[0], ptrs[1], ptrs[2], [ptrs[3], ptrs[4]])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    law.device_stats(0)
    return best


def zero_copy_torch(law, n, g, s0, a0, reps):
    pin = lambda k: torch.empty(k, dtype=torch.float64, pin_memory=True)  # noqa: E731
    tg, ts, tt, te, ta = pin(9 * n), pin(6 * n), pin(36 * n), pin(6 * n), pin(n)
    tg.numpy()[:] = g

    def reset():
        ts.numpy()[:] = s0
        te.zero_()
        ta.numpy()[:] = a0

    best = zero_copy_run(law, n, [x.data_ptr() for x in (tg, ts, tt, te, ta)], reset, reps)
    return best, (ts.numpy().copy(), tt.numpy().copy(), te.numpy().copy(), ta.numpy().copy())


def zero_copy_registered(law, n, g, s0, a0, reps):
    hip = hip_runtime()
    hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
    s, t, e, a = s0.copy(), np.zeros(36 * n), np.zeros(6 * n), a0.copy()
    arrays = [g, s, t, e, a]
    ctx = law._handle(0).ctx
    ptrs = []
    for x in arrays:
        ctx.register_host_buffer(x)
        d = C.c_void_p()
        rc = hip.hipHostGetDevicePointer(C.byref(d), C.c_void_p(x.ctypes.data), 0)
        assert rc == 0, rc
        ptrs.append(d.value)

    def reset():
        s[:] = s0
        e[:] = 0.0
        a[:] = a0

    try:
        best = zero_copy_run(law, n, ptrs, reset, reps)
    finally:
        for x in arrays:
            ctx.unregister_host_buffer(x)
    return best, (s, t, e, a), [hex(p) for p in ptrs[:2]], [hex(x.ctypes.data) for x in arrays[:2]]


def main():
    law = fc.VonMises3D(VM_P)
    for n in (1_000, 10_000, 100_000, 1_000_000, 10_000_000):
        reps = 4 if n >= 1_000_000 else 20
        g, s0, a0 = inputs(n)
        ta, ra = staged(law, n, g, s0, a0, reps)
        tb, rb = zero_copy_torch(law, n, g, s0, a0, reps)
        tc, rc, dptr, hptr = zero_copy_registered(law, n, g, s0, a0, reps)
        same_b = all(np.array_equal(x, y) for x, y in zip(ra, rb))
        same_c = all(np.array_equal(x, y) for x, y in zip(ra, rc))
        print(json.dumps({"n": n, "staged_ms": round(ta * 1e3, 3), "zero_copy_pinned_ms": round(tb * 1e3, 3),
                          "zero_copy_registered_ms": round(tc * 1e3, 3),
                          "staged_Mpts_s": round(n / ta / 1e6, 1), "zero_copy_pinned_Mpts_s": round(n / tb / 1e6, 1),
                          "zero_copy_registered_Mpts_s": round(n / tc / 1e6, 1),
                          "pcie_GBs_zero_copy": round(n * 568 / min(tb, tc) / 1e9, 1),
                          "bit_identical": [same_b, same_c], "dev_ptr": dptr, "host_ptr": hptr}), flush=True)


if __name__ == "__main__":
    main()
