#!/usr/bin/env python3
"""Threshold of the row-masked history access: kernel time of the sparse trial-history protocol against
the plastic fraction p of a random mixture and the `masked_max` context option (FCAMD_MASKED_MAX) (tiles with more touched rows take the
dense tile path; 0 = always dense, 64 = always masked).  One process, fixed arrays (fixed placement)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
f = dict(dtype=torch.float64, device=dev)
for wl in ("von_mises_mixed", "comfe_mises_mixed", "drucker_prager_mixed"):
    kind = bench.WORKLOADS[wl][0]
    law, _ = bench.make_law(kind)
    gen = torch.Generator(device=dev).manual_seed(3)
    g = torch.randn(9 * n, generator=gen, **f)
    u = torch.rand(n, generator=gen, **f)
    s0 = torch.zeros(6 * n, **f)
    if kind == "von_mises_3d":
        h0 = {"eps_n": torch.zeros(6 * n, **f), "alpha": torch.rand(n, generator=gen, **f) * 0.02}
    else:
        h0 = {"history": torch.zeros(7 * n, **f)}
        if kind == "comfe_drucker_prager":
            s0.view(n, 6)[:, :3] = -1000.0
    h1 = {k: v.clone() for k, v in h0.items()}
    s1, t = torch.empty_like(s0), torch.empty(36 * n, **f)
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    for p in (0.05, 0.1, 0.2, 0.35, 0.5, 0.75):
        gg = g.clone()
        gv = gg.view(n, 9)
        gv.mul_(torch.where(u < p, 1e-2, 1e-5).to(torch.float64)[:, None])
        if kind == "comfe_drucker_prager":  # mostly isochoric, as in bench.py
            gv.mul_(0.5)
            tr = (gv[:, 0] + gv[:, 4] + gv[:, 8]) * (0.95 / 3.0)
            for c in (0, 4, 8):
                gv[:, c] -= tr
        row = {"workload": wl, "p": p}
        for mm in (0, 8, 16, 24, 32, 48, 64):
            law._handle(0).ctx.set_option("masked_max", mm)
            for k in h0:
                h1[k].copy_(h0[k])
            mask.zero_()
            run = lambda: law.evaluate_from(0, 2.0, gg, s0, s1, t, h0, h1, history_mask=mask)  # noqa: E731
            run(), run()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
            for x, y in ev:
                x.record()
                run()
                y.record()
            torch.cuda.synchronize()
            row[f"mm{mm}"] = round(sum(x.elapsed_time(y) for x, y in ev) / 4, 3)
        row["plastic"] = round(law.device_stats().n_plastic / n, 3)
        print(json.dumps(row), flush=True)
        del gg, gv
    del g, u, s0, s1, t, h0, h1, mask, law
    torch.cuda.empty_cache()
