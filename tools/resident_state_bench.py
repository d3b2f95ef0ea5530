#!/usr/bin/env python3
"""The product object next to bench.py's headline: the SAME workload through `ResidentState.evaluate` (default
construction apart from sparse_tangent=False, so that every launch rewrites the tangent as the reference contract and
bench.py's headline do).  The first evaluate runs the state's placement step; the next ones are timed with events.
    python tools/resident_state_bench.py [n] [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fenics_constitutive_amd.resident import ResidentState  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
wl = bench.Workload("von_mises_mixed", n, 1234, dev, 0)  # the synthetic committed state and the two Newton iterates
st = ResidentState(wl.law, n, device=dev, stress0=wl.stress_c, history0=wl.hist_c, sparse_tangent=False)
grads = wl.grads
wl.stress_c = wl.stress_t = wl.hist_c = wl.hist_t = wl.tangent = wl.hmask = None
torch.cuda.empty_cache()
st.evaluate(0.0, wl.del_t, grads[0])  # placement happens here
for i in range(3):
    st.evaluate(0.0, wl.del_t, grads[i & 1])
n_pl = []
for i in (0, 1):
    st.evaluate(0.0, wl.del_t, grads[i])
    n_pl.append(int(st.check().n_plastic))
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
for i, (a, b) in enumerate(ev):
    a.record()
    st.evaluate(0.0, wl.del_t, grads[i & 1])
    b.record()
torch.cuda.synchronize()
st.update()  # reads the counters of the last evaluate, then swaps
ms = [a.elapsed_time(b) for a, b in ev]
avg = sum(ms) / len(ms)
npl = 0.5 * (n_pl[0] + n_pl[1])
alg = (n - npl) * 464 + npl * 568
print(json.dumps({"n": n, "steps": steps, "ResidentState_evaluate_ms_avg": round(avg, 4), "ms_min": round(min(ms), 4),
                  "frac": round(alg / (avg * 1e-3) / 1e9 / 8000.0, 4), "plastic_fraction": round(npl / n, 4),
                  "placement": {k: v for k, v in (st.placement or {}).items() if k != "arrays"}}))
