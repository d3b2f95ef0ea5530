#!/usr/bin/env python3
"""Slow-placement reproducer: all arrays carved from one 80 GiB allocation (consistently ~20 % slower
than separate allocations).  A/B launch knobs on it (env re-read per call)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc

n = 100_000_000
law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
dev = torch.device("cuda", 0)
slab = torch.empty(80 * (1 << 30), dtype=torch.uint8, device=dev)
sizes = [9 * n, 6 * n, 6 * n, 6 * n, 6 * n, n, n, 36 * n]
arrs, off = [], 0
for m in sizes:
    off = -(-off // (1 << 21)) * (1 << 21)
    arrs.append(slab[off : off + 8 * m].view(torch.float64))
    off += 8 * m
g, s0, s1, e0, e1, a0, a1, t = arrs
gen = torch.Generator(device=dev).manual_seed(1)
g.normal_(generator=gen)
g.view(n, 9).mul_(torch.pow(10.0, torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 4)[:, None])
s0.zero_(), e0.zero_(), a0.uniform_(0, 0.02, generator=gen)
h0, h1 = {"eps_n": e0, "alpha": a0}, {"eps_n": e1, "alpha": a1}
variants = [v.split("=") for v in sys.argv[1:]]
res = {tuple(v): [] for v in variants}
for rnd in range(4):
    for k, v in variants:
        if k == "GRID":
            law._handle(0).ctx.set_grid(int(v))
        else:  # FCAMD_TILE_MAP / FCAMD_NT / FCAMD_MASKED_MAX: context options since round 2
            law._handle(0).ctx.set_option({"FCAMD_TILE_MAP": "tile_map", "FCAMD_NT": "nontemporal", "FCAMD_MASKED_MAX": "masked_max"}[k], int(v))
        for _ in range(2):
            law.evaluate_from(0, 1, g, s0, s1, t, h0, h1)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record(); law.evaluate_from(0, 1, g, s0, s1, t, h0, h1); b.record()
        torch.cuda.synchronize()
        res[(k, v)].append(sum(a.elapsed_time(b) for a, b in ev) / len(ev))
for k, v in res.items():
    print(k, "median %.3f ms" % sorted(v)[len(v) // 2], ["%.2f" % x for x in v], flush=True)
