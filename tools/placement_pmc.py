#!/usr/bin/env python3
"""VonMises3D mixed, 1e8 points, arrays either carved from one 80 GiB allocation ("slab": the
reproducible slow placement) or allocated separately; a handful of launches, meant to be run under
`rocprofv3 --pmc ...` to compare TLB / DRAM-stall counters between the two placements."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc

mode = sys.argv[1]
n = 100_000_000
law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
dev = torch.device("cuda", 0)
sizes = [9 * n, 6 * n, 6 * n, 6 * n, 6 * n, n, n, 36 * n]
if mode == "slab":
    slab = torch.empty(80 * (1 << 30), dtype=torch.uint8, device=dev)
    arrs, off = [], 0
    for m in sizes:
        off = -(-off // (1 << 21)) * (1 << 21)
        arrs.append(slab[off : off + 8 * m].view(torch.float64))
        off += 8 * m
else:
    arrs = [torch.empty(m, dtype=torch.float64, device=dev) for m in sizes]
g, s0, s1, e0, e1, a0, a1, t = arrs
gen = torch.Generator(device=dev).manual_seed(1)
g.normal_(generator=gen)
g.view(n, 9).mul_(torch.pow(10.0, torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 4)[:, None])
s0.zero_(), e0.zero_(), a0.uniform_(0, 0.02, generator=gen)
h0, h1 = {"eps_n": e0, "alpha": a0}, {"eps_n": e1, "alpha": a1}
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
for a, b in ev:
    a.record(); law.evaluate_from(0, 1, g, s0, s1, t, h0, h1); b.record()
torch.cuda.synchronize()
print(mode, "kernel ms", ["%.3f" % a.elapsed_time(b) for a, b in ev], flush=True)
