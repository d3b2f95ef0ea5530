#!/usr/bin/env python3
"""VonMises3D kernel time vs launch grid at small/medium n (diagnostic)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc

law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, fc.StressStrainConstraint.FULL)
dev = torch.device("cuda", 0)
f = dict(dtype=torch.float64, device=dev)
for n in (100_000, 1_000_000, 10_000_000):
    gen = torch.Generator(device=dev).manual_seed(1)
    g = torch.randn(9 * n, generator=gen, **f)
    g.view(n, 9).mul_(torch.pow(10.0, torch.rand(n, generator=gen, **f) * 2 - 4)[:, None])
    s0, s1 = torch.zeros(6 * n, **f), torch.empty(6 * n, **f)
    h0 = {"eps_n": torch.zeros(6 * n, **f), "alpha": torch.rand(n, generator=gen, **f) * 0.02}
    h1 = {k: torch.empty_like(v) for k, v in h0.items()}
    t = torch.empty(36 * n, **f)
    for name, fn, lw in (("vm", lambda: law.evaluate_from(0, 1, g, s0, s1, t, h0, h1), law), ("le", lambda: le.evaluate_from(0, 1, g, s0, s1, t, None, None), le)):
        res = []
        for grid in (256, 512, 1024, 2048, 4096, 16384):
            lw._handle(0).ctx.set_grid(grid)
            for _ in range(3):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in ev:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ev)
            res.append(f"g{grid}:{ms[len(ms)//2]*1e3:.0f}us")
        print(f"n={n} {name}: " + "  ".join(res), flush=True)
