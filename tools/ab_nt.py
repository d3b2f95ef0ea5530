#!/usr/bin/env python3
"""In-process A/B of non-temporal vs plain global accesses for every law at n = 1e8."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
n = 100_000_000
for wl in ("linear_elasticity", "spring_maxwell", "spring_kelvin", "von_mises_mixed", "comfe_mises_mixed"):
    kind, scale, _, _ = bench.WORKLOADS[wl]
    law, _ = bench.make_law(kind)
    grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1, dev)
    g = grad_array()
    t = torch.empty(36 * n, dtype=torch.float64, device=dev)
    s1 = torch.empty_like(s0)
    h1 = None if h0 is None else {k: torch.empty_like(v) for k, v in h0.items()}
    res = {"1": [], "0": []}
    for rnd in range(4):
        for nt in ("1", "0"):
            law._handle(0).ctx.set_option("nontemporal", int(nt))
            for _ in range(2):
                law.evaluate_from(0, 2.0, g, s0, s1, t, h0, h1)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in ev:
                a.record(); law.evaluate_from(0, 2.0, g, s0, s1, t, h0, h1); b.record()
            torch.cuda.synchronize()
            res[nt].append(sum(a.elapsed_time(b) for a, b in ev) / len(ev))
    print(wl, "nt=1 %.3f ms" % sorted(res["1"])[2], "nt=0 %.3f ms" % sorted(res["0"])[2], flush=True)
    del g, t, s0, s1, h0, h1, law
    torch.cuda.empty_cache()
