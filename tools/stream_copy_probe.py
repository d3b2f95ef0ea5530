#!/usr/bin/env python3
"""fcamd_copy_device across launch grids and buffer pairs: what this box's memory gives a plain 16-byte-per-lane
non-temporal copy (the "achievable" figure next to the 8 TB/s peak).    python tools/stream_copy_probe.py [GiB per buffer]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenics_constitutive_amd import _capi, hostio  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(gib * (1 << 30)) // 8
ctx = _capi.get_context(0)


def rate(dst, src, reps=5):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    hostio.copy_device(dst, src)
    for a, b in ev:
        a.record()
        hostio.copy_device(dst, src)
        b.record()
    torch.cuda.synchronize()
    return round(2 * 8 * n / (min(a.elapsed_time(b) for a, b in ev) * 1e-3) / 1e9, 1)


pairs = [(torch.empty(n, dtype=torch.float64, device="cuda"), torch.ones(n, dtype=torch.float64, device="cuda")) for _ in range(3)]
for grid in (0, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 262144):
    ctx.set_grid(grid)
    print(json.dumps({"grid": grid or "default (32 per CU)", "GBs_per_pair": [rate(d, s) for d, s in pairs],
                      "torch_copy_GBs": None}), flush=True)
ctx.set_grid(0)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
for d, s in pairs:
    for a, b in ev:
        a.record()
        d.copy_(s)
        b.record()
    torch.cuda.synchronize()
    print(json.dumps({"torch_copy_GBs": round(2 * 8 * n / (min(a.elapsed_time(b) for a, b in ev) * 1e-3) / 1e9, 1)}), flush=True)
