#!/usr/bin/env python3
"""Does kernel time depend on where the buffers land?  Re-allocate all arrays several times in one
process and time the same VonMises3D launch (diagnostic for run-to-run variance)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
law = fc.VonMises3D({"p_ka": 175000.0, "p_mu": 80769.0, "p_y0": 1200.0, "p_y00": 2500.0, "p_w": 200.0})
dev = torch.device("cuda", 0)


def trial(tag, pad_bytes=0):
    gen = torch.Generator(device=dev).manual_seed(1)
    pads = []

    def alloc(m):
        if pad_bytes:
            pads.append(torch.empty(pad_bytes, dtype=torch.uint8, device=dev))
        return torch.empty(m, dtype=torch.float64, device=dev)

    g = alloc(9 * n)
    g.normal_(generator=gen)
    g.view(n, 9).mul_(torch.pow(10.0, torch.rand(n, dtype=torch.float64, device=dev, generator=gen) * 2 - 4)[:, None])
    s0, s1 = alloc(6 * n).zero_(), alloc(6 * n)
    e0, e1 = alloc(6 * n).zero_(), alloc(6 * n)
    a0, a1 = alloc(n).uniform_(0, 0.02, generator=gen), alloc(n)
    t = alloc(36 * n)
    h0, h1 = {"eps_n": e0, "alpha": a0}, {"eps_n": e1, "alpha": a1}
    for _ in range(3):
        law.evaluate_from(0, 1, g, s0, s1, t, h0, h1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record()
        law.evaluate_from(0, 1, g, s0, s1, t, h0, h1)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    ptrs = [x.data_ptr() for x in (g, s0, s1, e0, e1, a0, a1, t)]
    print(f"{tag}: avg {sum(ms)/len(ms):.3f} ms min {ms[0]:.3f} max {ms[-1]:.3f}  ptr(MiB) {[p >> 20 for p in ptrs]}", flush=True)


import random
random.seed(0)
for i in range(3):
    trial(f"realloc{i}")
    torch.cuda.empty_cache()
for i in range(12):
    pad = random.randrange(1, 4000) * (1 << 21)
    trial(f"pad{pad >> 20}MiB", pad)
    torch.cuda.empty_cache()
