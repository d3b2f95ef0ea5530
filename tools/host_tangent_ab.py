"""Interleaved A/B of host-tangent settings in one process (the box's host cores are shared with other tenants: single runs differ by
+-20 %): python tools/host_tangent_ab.py N "threads=16,chunk=0" "threads=16,chunk=1048576" ..."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd import _capi  # noqa: E402

n = int(float(sys.argv[1]))
cfgs = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[2:]]
rng = np.random.default_rng(3)
ctx = _capi.get_context(_capi.default_device())
LAW = os.environ.get("AB_LAW", "vm")  # vm | le | dp
if LAW == "vm":
    law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
elif LAW == "dp":
    law = fc.DruckerPragerHyperbolic3D({k: np.array([v]) for k, v in {"mu": 80769.0, "kappa": 175000.0, "a": 100.0, "b": 0.05, "b_flow": 0.02, "d": 40.0}.items()})
else:
    law = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL)
g = rng.standard_normal(9 * n)
g *= np.repeat(10.0 ** (rng.random(n) * 2.0 - 4.0), 9)
a0 = rng.random(n) * 0.02
s, t, e, a = np.zeros(6 * n), np.zeros(36 * n), np.zeros(6 * n), a0.copy()
h7 = np.zeros(7 * n)
if LAW == "dp":  # compressive prestress and mostly isochoric increments (the regime in which the reference's Newton iteration converges)
    g *= 0.1
    gv = g.reshape(-1, 9)
    gv[:, [0, 4, 8]] -= (0.95 * gv[:, [0, 4, 8]].sum(axis=1) / 3.0)[:, None]
for x in (g, s, t, e, a, h7):
    ctx.register_host_buffer(x)
times = [[] for _ in cfgs]
cpu = [[] for _ in cfgs]
for rnd in range(int(os.environ.get("AB_ROUNDS", "8"))):
    for k, c in enumerate(cfgs):
        ctx.set_option("host_tangent_threads", int(c.get("threads", -1)))
        ctx.set_option("host_tangent_chunk", int(c.get("chunk", 0)))
        ctx.set_option("host_tangent_streams", int(c.get("streams", 1)))
        ctx.set_option("host_tangent_min_points", int(c.get("min", 65536)))
        s[:] = 0.0
        if LAW == "dp":
            s.reshape(-1, 6)[:, :3] = -1000.0
            h7[:] = 0.0
        e[:] = 0.0
        a[:] = a0
        t0 = time.perf_counter()
        law.evaluate(0.0, 1.0, g, s, t, {"eps_n": e, "alpha": a} if LAW == "vm" else ({"history": h7} if LAW == "dp" else None))
        dt = time.perf_counter() - t0
        if rnd >= 2:
            times[k].append(dt)
            cpu[k].append(ctx.get_option("last_host_tangent_cpu_us") / 1e3)
for k, c in enumerate(cfgs):
    ts = sorted(times[k])
    print(json.dumps({"n": n, **c, "median_ms": round(ts[len(ts) // 2] * 1e3, 2), "min_ms": round(ts[0] * 1e3, 2), "Mpts_s_median": round(n / ts[len(ts) // 2] / 1e6, 1),
                      "Mpts_s_best": round(n / ts[0] / 1e6, 1), "cpu_ms_median": round(sorted(cpu[k])[len(cpu[k]) // 2], 1)}), flush=True)
