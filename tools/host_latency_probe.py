#!/usr/bin/env python3
"""Per-call time of the two host entries at small sizes: ndarray evaluate (fcamd_evaluate_host) and
ResidentState.evaluate_into (fcamd_evaluate_resident), pageable vs registered arrays.  Run a second
time with FCAMD_ZERO_COPY=0 for the staged-but-page-locked numbers."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402

VM_P = {"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0}


def own(a):
    import mmap

    out = np.frombuffer(mmap.mmap(-1, max(a.nbytes, 8)), dtype=np.float64, count=a.size)
    out[:] = a
    return out


def main():
    law = fc.VonMises3D(VM_P)
    ctx = law._handle(0).ctx
    for n in (1_000, 10_000, 100_000, 1_000_000):
        rng = np.random.default_rng(0)
        g = own(rng.normal(size=9 * n) * np.repeat(10 ** rng.uniform(-4, -2, size=n), 9))
        s, t, e, a = own(np.zeros(6 * n)), own(np.zeros(36 * n)), own(np.zeros(6 * n)), own(rng.uniform(0, 0.02, size=n))
        st = ResidentState(law, n, history0={"eps_n": e, "alpha": a})
        row = {"n": n}
        for reg in (False, True):
            if reg:
                for x in (g, s, t, e, a):
                    ctx.register_host_buffer(x)
            reps = 50 if n <= 100_000 else 8
            for name, fn in (("evaluate", lambda: law.evaluate(0.0, 1.0, g, s, t, {"eps_n": e, "alpha": a})),
                             ("evaluate_into", lambda: st.evaluate_into(0.0, 1.0, g, s, t))):
                best = None
                for _ in range(reps):
                    s[:] = 0.0
                    e[:] = 0.0
                    t0 = time.perf_counter()
                    fn()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                row[f"{name}_{'registered' if reg else 'pageable'}_us"] = round(best * 1e6, 1)
                row[f"{name}_{'registered' if reg else 'pageable'}_mode"] = ctx.last_host_mode()
            if reg:
                for x in (g, s, t, e, a):
                    ctx.unregister_host_buffer(x)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
