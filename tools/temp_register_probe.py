#!/usr/bin/env python3
"""Host (ndarray) entry on arrays the library has never seen: the runtime's pageable-copy path (hipMemcpyAsync on the
caller's memory: page-locks it piecewise on the fly and CACHES the locks) against page-locking the arrays for the
duration of the call (register -> zero-copy launch -> unregister).      python tools/temp_register_probe.py [n ...]"""
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd import _capi  # noqa: E402

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}
law = fc.VonMises3D(VM_P)
law.auto_pin = False if hasattr(law, "auto_pin") else None
ctx = law._handle(_capi.default_device()).ctx
for n in [int(float(a)) for a in sys.argv[1:]] or [100_000, 1_000_000, 10_000_000]:
    rng = np.random.default_rng(0)
    g = rng.normal(size=9 * n) * 3e-3

    def fresh():
        return [g.copy(), np.zeros(6 * n), np.empty(36 * n), np.zeros(6 * n), np.zeros(n)]

    rec = {"n": n}
    for mode in ("pageable", "temp_register", "pageable", "temp_register"):
        arrs = fresh()
        t0 = time.perf_counter()
        if mode == "temp_register":
            for a in arrs:
                ctx.register_host_buffer(a)
            t1 = time.perf_counter()
        law.evaluate(0.0, 1.0, arrs[0], arrs[1], arrs[2], {"eps_n": arrs[3], "alpha": arrs[4]})
        t2 = time.perf_counter()
        if mode == "temp_register":
            for a in arrs:
                ctx.unregister_host_buffer(a)
            t3 = time.perf_counter()
            rec.setdefault("temp_register_ms", []).append({"register": round(1e3 * (t1 - t0), 2), "evaluate": round(1e3 * (t2 - t1), 2),
                                                           "unregister": round(1e3 * (t3 - t2), 2), "total": round(1e3 * (t3 - t0), 2),
                                                           "host_mode": ctx.last_host_mode()})
        else:
            rec.setdefault("pageable_ms", []).append({"total": round(1e3 * (t2 - t0), 2), "host_mode": ctx.last_host_mode()})
        del arrs
    # the same arrays call after call (a solver's persistent arrays)
    for mode in ("pageable", "temp_register"):
        arrs = fresh()
        for rep in range(4):
            t0 = time.perf_counter()
            if mode == "temp_register":
                for a in arrs:
                    ctx.register_host_buffer(a)
            law.evaluate(0.0, 1.0, arrs[0], arrs[1], arrs[2], {"eps_n": arrs[3], "alpha": arrs[4]})
            if mode == "temp_register":
                for a in arrs:
                    ctx.unregister_host_buffer(a)
            rec.setdefault("reused_" + mode + "_ms", []).append(round(1e3 * (time.perf_counter() - t0), 2))
        del arrs
    print(json.dumps(rec), flush=True)
