#!/usr/bin/env python3
"""In-process A/B of launch knobs (context options) on identical buffers: interleaved rounds.
    python tools/ab_probe.py FCAMD_MASKED_MAX=0 FCAMD_MASKED_MAX=20 ...      (AB_SPARSE=1: every variant with the sparse trial-history mask)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402

n = 100_000_000
law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
f = dict(dtype=torch.float64, device=dev)
g = torch.randn(9 * n, generator=gen, **f)
g.view(n, 9).mul_(torch.pow(10.0, torch.rand(n, generator=gen, **f) * 2 - 4)[:, None])
s0, s1, e0, e1 = torch.zeros(6 * n, **f), torch.empty(6 * n, **f), torch.zeros(6 * n, **f), torch.empty(6 * n, **f)
a0, a1, t = torch.rand(n, generator=gen, **f) * 0.02, torch.empty(n, **f), torch.empty(36 * n, **f)
h0, h1 = {"eps_n": e0, "alpha": a0}, {"eps_n": e1, "alpha": a1}
mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
e1.copy_(e0), a1.copy_(a0)  # sparse contract: trial == committed
# launch knobs are context options (fcamd_context_set_option); the FCAMD_* environment names select them here
variants = [v.split("=") for v in sys.argv[1:]] or [["FCAMD_TILE_MAP", "0"], ["FCAMD_TILE_MAP", "1"]]
OPTION = {"FCAMD_TILE_MAP": "tile_map", "FCAMD_MASKED_MAX": "masked_max"}
ctx = law._handle(0).ctx
res = {tuple(v): [] for v in variants}
for rnd in range(6):
    for k, v in variants:
        if k in OPTION:
            ctx.set_option(OPTION[k], int(v))
        hm = mask if ((k == "SPARSE" and v == "1") or os.environ.get("AB_SPARSE") == "1") else None
        for _ in range(2):
            law.evaluate_from(0, 1, g, s0, s1, t, h0, h1, history_mask=hm)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record()
            law.evaluate_from(0, 1, g, s0, s1, t, h0, h1, history_mask=hm)
            b.record()
        torch.cuda.synchronize()
        res[(k, v)].append(sum(a.elapsed_time(b) for a, b in ev) / len(ev))
for k, v in res.items():
    print(k, "median %.3f ms" % sorted(v)[len(v) // 2], "min %.3f" % min(v), ["%.2f" % x for x in v])
