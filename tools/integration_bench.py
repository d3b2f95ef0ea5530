#!/usr/bin/env python3
"""Wall time of one Newton iteration's constitutive part (IncrSmallStrainProblem.form without dolfinx's own
work) for a two-material problem, driven with the stand-ins of tests/test_gpu_integration.py that restate
the reference's host protocol (state copies, map_to_sub, evaluate, map_to_parent):
  reference protocol + GPU laws on ndarrays   (law.evaluate per LawOnSubMesh, host maps)
  use_resident_state                          (per-law device state; host map_to_parent stays)
  use_resident_problem_state                  (one device state; kernels write the global arrays)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_integration as T  # noqa: E402
from fenics_constitutive_amd.integration import use_resident_problem_state, use_resident_state  # noqa: E402

n_cells = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000
q = 4


def run(mode):
    p = T.build(n_cells, q, 3)
    if mode == "resident per law":
        use_resident_state(p)
    elif mode == "resident problem state":
        use_resident_problem_state(p)
    else:
        for los in p._law_on_submeshs:
            los.law.auto_pin = True
    rng = np.random.default_rng(0)
    grads = []
    for los in p._law_on_submeshs:
        m = los.stress.x.array.size // 6
        grads.append(rng.normal(size=9 * m) * np.repeat(10 ** rng.uniform(-4, -2, size=m), 9))
    times = []
    for it in range(6):
        for los, g in zip(p._law_on_submeshs, grads):
            p.incr_disp.grads[los.cells.tobytes()] = g * (1.0 + 0.01 * it)
        t0 = time.perf_counter()
        p.form()
        times.append(time.perf_counter() - t0)
        if it == 2:
            p.update()
    best = min(times[1:])
    for los in p._law_on_submeshs:
        los.law.unpin_arrays()
    return {"mode": mode, "points": n_cells * q, "form_ms": round(best * 1e3, 2), "Mpts_s": round(n_cells * q / best / 1e6, 1),
            "checksum": float(p.stress.current.x.array.sum())}


for mode in ("reference protocol, GPU laws on ndarrays", "resident per law", "resident problem state"):
    print(json.dumps(run(mode)), flush=True)


def run_single(mode):
    """One material on the whole mesh (IdentityMap: the maps are plain copies)."""
    p = T.build_single(n_cells, q, 3)
    los = p._law_on_submeshs[0]
    if mode != "reference protocol":
        use_resident_state(p, direct_global=(mode == "resident, direct to the global arrays"))
    else:
        los.law.auto_pin = True
    rng = np.random.default_rng(0)
    m = n_cells * q
    g = rng.normal(size=9 * m) * np.repeat(10 ** rng.uniform(-4, -2, size=m), 9)
    times = []
    for it in range(6):
        p.incr_disp.grads[los.cells.tobytes()] = g * (1.0 + 0.01 * it)
        t0 = time.perf_counter()
        p.form()
        times.append(time.perf_counter() - t0)
        if it == 2:
            p.update()
    best = min(times[1:])
    los.law.unpin_arrays()
    return {"mode": "single material: " + mode, "points": m, "form_ms": round(best * 1e3, 2), "Mpts_s": round(m / best / 1e6, 1),
            "checksum": float(p.stress.current.x.array.sum())}


for mode in ("reference protocol", "resident, local arrays + map_to_parent", "resident, direct to the global arrays"):
    print(json.dumps(run_single(mode)), flush=True)
