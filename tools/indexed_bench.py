#!/usr/bin/env python3
"""Throughput of the submesh-indexed evaluate vs gather + evaluate + scatter with separate map kernels."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc
from fenics_constitutive_amd.maps import DeviceSubSpaceMap

dev = torch.device("cuda", 0)
f = dict(dtype=torch.float64, device=dev)
law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
n_parent, n_sub = 60_000_000, 30_000_000


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


gen = torch.Generator(device=dev).manual_seed(3)
g = torch.randn(9 * n_sub, generator=gen, **f)
g.view(n_sub, 9).mul_(torch.pow(10.0, torch.rand(n_sub, generator=gen, **f) * 2 - 4)[:, None])
h0 = {"eps_n": torch.zeros(6 * n_sub, **f), "alpha": torch.rand(n_sub, generator=gen, **f) * 0.02}
h1 = {k: torch.empty_like(v) for k, v in h0.items()}
sp, sc, tp = torch.zeros(6 * n_parent, **f), torch.zeros(6 * n_parent, **f), torch.zeros(36 * n_parent, **f)
s_loc, t_loc = torch.zeros(6 * n_sub, **f), torch.zeros(36 * n_sub, **f)
for order in ("contiguous block", "sorted random half", "random order"):
    if order == "contiguous block":
        rows = torch.arange(n_sub, device=dev) + 1000
    else:
        rows = torch.randperm(n_parent, device=dev)[:n_sub]
        if order.startswith("sorted"):
            rows = rows.sort().values
    rows = rows.to(torch.int32).contiguous()
    m = DeviceSubSpaceMap(rows, torch.arange(n_sub, dtype=torch.int32, device=dev), device=dev)
    fused = timeit(lambda: law.evaluate_indexed(0, 1, g, sp, sc, tp, rows, h0, h1))

    def unfused():
        m.map_to_sub(sp, s_loc, 6)                       # gather committed stress
        law.evaluate_from(0, 1, g, s_loc, s_loc, t_loc, h0, h1)
        m.map_to_parent(s_loc, sc, 6)                    # scatter stress and tangent
        m.map_to_parent(t_loc, tp, 36)

    sep = timeit(unfused)
    print(json.dumps({"rows": order, "n_sub": n_sub, "fused_ms": round(fused, 3), "fused_Gpts_s": round(n_sub / fused / 1e6, 2),
                      "separate_maps_ms": round(sep, 3), "separate_Gpts_s": round(n_sub / sep / 1e6, 2)}), flush=True)
