#!/usr/bin/env python3
"""In-process A/B of launch knobs (context options) on bench.py's headline workload -- identical buffers, interleaved rounds:
    python tools/ab_knobs.py tile_map=0 tile_map=1 grid=8192 grid=32768 ...       (AB_WORKLOAD=<name> | AB_FROW=<row>, AB_POINTS=<n>)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(float(os.environ.get("AB_POINTS", "1e8")))
if os.environ.get("AB_FROW"):  # a SURVEY 8(f) row (benchlib/frows.py) instead of a BASELINE workload
    from benchlib import frows

    row = frows.FROWS[os.environ["AB_FROW"]](n, torch.device("cuda", 0))

    class _W:
        law = row.law

        def launch(self, i):
            row.reset()
            row.launch()

        def timed_events(self, k):
            return row.timed(k)

    wl = _W()
else:
    wl = bench.Workload(os.environ.get("AB_WORKLOAD", bench.HEADLINE), n, 1234, torch.device("cuda", 0), 0)
ctx = wl.law._handle(0).ctx
defaults = {"tile_map": 0, "grid": 0, "masked_max": -1}
variants = [v.split("=") for v in sys.argv[1:]] or [["tile_map", "0"], ["tile_map", "1"]]
res = {tuple(v): [] for v in variants}
for rnd in range(5):
    for k, v in variants:
        for kk, dv in defaults.items():
            ctx.set_option(kk, dv)
        ctx.set_option(k, int(v))
        wl.launch(0), wl.launch(1)
        ms = wl.timed_events(6)
        res[(k, v)].append(sum(ms) / len(ms))
for k, v in res.items():
    print(k, "median %.3f ms" % sorted(v)[len(v) // 2], "min %.3f" % min(v), ["%.2f" % x for x in v], flush=True)
