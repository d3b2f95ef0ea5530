#!/usr/bin/env python3
"""Do launch knobs rescue a slow tangent placement?  Candidate allocations of the tangent alone (all else
fixed, as tools/tangent_placement_probe.py), kernel time per candidate under: default; plain (temporal)
accesses FCAMD_NT=0; XCD-contiguous tile map FCAMD_TILE_MAP=1; smaller grids (fewer concurrent tiles)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = sys.argv[3] if len(sys.argv) > 3 else "von_mises_mixed"
kind, scale, _, _ = bench.WORKLOADS[wl]
law, _ = bench.make_law(kind)
grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1234, dev)
t0 = torch.empty(36 * n, dtype=torch.float64, device=dev)
gw = grad_array()
law.evaluate(0, 2.0, gw, s0, t0, h0)
del gw
g = grad_array()
s1 = torch.empty_like(s0)
h1 = None if h0 is None else {kk: torch.empty_like(v) for kk, v in h0.items()}
ctx = law._handle(0).ctx


def ms_of(t, reps=4):
    fn = lambda: law.evaluate_from(0, 2.0, g, s0, s1, t, h0, h1)  # noqa: E731
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in ev:
        x.record()
        fn()
        y.record()
    torch.cuda.synchronize()
    return round(min(x.elapsed_time(y) for x, y in ev), 3)


cands = [t0] + [torch.empty(36 * n, dtype=torch.float64, device=dev) for _ in range(k - 1)]
for i, t in enumerate(cands):
    row = {"cand": i}
    ctx.set_option("nontemporal", 1), ctx.set_option("tile_map", 0)
    ctx.set_grid(0)
    row["default"] = ms_of(t)
    ctx.set_option("nontemporal", 0)
    row["nt0"] = ms_of(t)
    ctx.set_option("nontemporal", 1)
    ctx.set_option("tile_map", 1)
    row["xcd_map"] = ms_of(t)
    ctx.set_option("tile_map", 0)
    for grid in (256, 512, 1024, 2048, 4096):
        ctx.set_grid(grid)
        row[f"grid{grid}"] = ms_of(t)
    ctx.set_grid(0)
    print(json.dumps(row), flush=True)
