#!/usr/bin/env python3
"""Per-call time of ResidentState.evaluate issued eagerly vs replayed from a captured HIP graph (device
gradients).  Negative result kept for the record: one kernel per call leaves a graph nothing to amortise --
19 / 19 / 21 / 60 us eager vs 26 / 26 / 28 / 66 us per replay at 1e3 / 1e4 / 1e5 / 1e6 points."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402

law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
rng = np.random.default_rng(0)
for n in (1_000, 10_000, 100_000, 1_000_000):
    g = torch.from_numpy(rng.normal(size=9 * n) * np.repeat(10 ** rng.uniform(-4, -2, size=n), 9)).cuda()
    st = ResidentState(law, n, history0={"eps_n": np.zeros(6 * n), "alpha": rng.uniform(0, 0.02, size=n)})
    st.grad.copy_(g)
    out = {"n": n}
    st.evaluate(0.0, 1.0, st.grad)
    st.evaluate(0.0, 1.0, st.grad)   # steady state of the tangent protocols before the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st.evaluate(0.0, 1.0, st.grad)
    for name, fn in (("eager", lambda: st.evaluate(0.0, 1.0, st.grad)), ("graph", graph.replay)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        reps = 500
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out[name + "_us"] = round((time.perf_counter() - t0) / reps * 1e6, 2)
    print(json.dumps(out), flush=True)
