#!/usr/bin/env python3
"""Is a slow tangent placement slow for every law?  k candidate allocations of the tangent, all other
arrays fixed per law; kernel time of each law on each candidate (tools/tangent_placement_probe.py for one law)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cands = [torch.empty(36 * n, dtype=torch.float64, device=dev) for _ in range(k)]
rows = [{"cand": i} for i in range(k)]
for wl in ("linear_elasticity", "spring_maxwell", "von_mises_mixed", "von_mises_plastic", "drucker_prager_mixed"):
    kind, scale, _, _ = bench.WORKLOADS[wl]
    law, _ = bench.make_law(kind)
    grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1234, dev)
    gw = grad_array()
    law.evaluate(0, 2.0, gw, s0, cands[0], h0)
    del gw
    g = grad_array()
    s1 = torch.empty_like(s0)
    h1 = None if h0 is None else {kk: torch.empty_like(v) for kk, v in h0.items()}
    for i, t in enumerate(cands):
        fn = lambda: law.evaluate_from(0, 2.0, g, s0, s1, t, h0, h1)  # noqa: E731
        fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
        for x, y in ev:
            x.record()
            fn()
            y.record()
        torch.cuda.synchronize()
        rows[i][wl] = round(min(x.elapsed_time(y) for x, y in ev), 3)
    del g, s0, s1, h0, h1, law
    torch.cuda.empty_cache()
for r in rows:
    print(json.dumps(r), flush=True)
