#!/usr/bin/env python3
"""The run-to-run variance of the evaluate kernel follows the placement of the WRITTEN arrays (above all
the tangent, 51-63 % of the traffic): tools/placement_retry_probe.py.  Here: k candidate allocations of
the tangent alone, everything else fixed -- kernel time per candidate next to cheap predictors measured
on the candidate itself (fill = pure write, copy of one half onto the other, sum = pure read)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kind, scale, _, _ = bench.WORKLOADS["von_mises_mixed"]
law, _ = bench.make_law(kind)
grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1234, dev)
t0 = torch.empty(36 * n, dtype=torch.float64, device=dev)
gw = grad_array()
law.evaluate(0, 2.0, gw, s0, t0, h0)
del gw
g = grad_array()
s1, e1, a1 = torch.empty_like(s0), torch.empty_like(h0["eps_n"]), torch.empty_like(h0["alpha"])


def ms_of(fn, reps=4):
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in ev:
        x.record()
        fn()
        y.record()
    torch.cuda.synchronize()
    return min(x.elapsed_time(y) for x, y in ev)


cands = [t0] + [torch.empty(36 * n, dtype=torch.float64, device=dev) for _ in range(k - 1)]
half = (36 * n // 2) & ~1
for i, t in enumerate(cands):
    kern = ms_of(lambda: law.evaluate_from(0, 2.0, g, s0, s1, t, h0, {"eps_n": e1, "alpha": a1}))
    fill = ms_of(lambda: t.fill_(1.0))
    cp = ms_of(lambda: t[:half].copy_(t[half:2 * half]))
    rd = ms_of(lambda: t.sum())
    print(json.dumps({"cand": i, "kernel_ms": round(kern, 3), "fill_TBs": round(8 * 36 * n / fill / 1e9, 3),
                      "copy_TBs": round(2 * 8 * half / cp / 1e9, 3), "sum_TBs": round(8 * 36 * n / rd / 1e9, 3),
                      "ptr": hex(t.data_ptr())}), flush=True)
