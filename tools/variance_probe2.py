#!/usr/bin/env python3
"""Kernel time vs relative placement of the arrays inside ONE slab (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402

n = 100_000_000
law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
dev = torch.device("cuda", 0)
slab = torch.empty(80 * (1 << 30), dtype=torch.uint8, device=dev)
base = slab.data_ptr()
assert base % (1 << 21) == 0
sizes = [9 * n, 6 * n, 6 * n, 6 * n, 6 * n, n, n, 36 * n]  # g s0 s1 e0 e1 a0 a1 t


def carve(align, stagger):
    """arrays placed one after another, each start rounded up to `align`, plus k*stagger bytes"""
    out, off = [], 0
    for k, m in enumerate(sizes):
        off = -(-off // align) * align + k * stagger
        out.append(slab[off : off + 8 * m].view(torch.float64))
        off += 8 * m
    assert off <= slab.numel()
    return out


def trial(tag, align, stagger):
    g, s0, s1, e0, e1, a0, a1, t = carve(align, stagger)
    gen = torch.Generator(device=dev).manual_seed(1)
    g.normal_(generator=gen)
    g.view(n, 9).mul_(torch.pow(10.0, torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 4)[:, None])
    s0.zero_(), e0.zero_(), a0.uniform_(0, 0.02, generator=gen)
    h0, h1 = {"eps_n": e0, "alpha": a0}, {"eps_n": e1, "alpha": a1}
    res = []
    for grid in (512, 1024, 2048, 4096, 16384):
        law._handle(0).ctx.set_grid(grid)
        for _ in range(2):
            law.evaluate_from(0, 1, g, s0, s1, t, h0, h1)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
        for a, b in ev:
            a.record()
            law.evaluate_from(0, 1, g, s0, s1, t, h0, h1)
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        res.append(f"g{grid}:{sum(ms)/len(ms):.2f}")
    print(f"{tag:28s} " + "  ".join(res), flush=True)


trial("slab contiguous", 16, 0)
trial("slab align 2MiB", 1 << 21, 0)
del slab
torch.cuda.empty_cache()


def carve(align, stagger):  # separate allocations instead of the slab
    return [torch.empty(m, dtype=torch.float64, device=dev) for m in sizes]


for i in range(4):
    trial(f"separate allocs {i}", 0, 0)
    torch.cuda.empty_cache()
