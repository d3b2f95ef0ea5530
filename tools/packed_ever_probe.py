#!/usr/bin/env python3
"""What the packed plastic-strain layout costs when EVER >> touched.  bench.py's headline workload (VonMises3D, 1e8
points), the packed protocol and the sparse protocol on the reference's layout on the SAME stress / gradient / tangent arrays of one
process (only the history arrays differ), interleaved rounds -- with the workload's own EVER set and with every row made non-zero
(EVER = all points), at the workload's plastic fraction and at a gradient scaled down to a few per cent of plastic points."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fenics_constitutive_amd.device import pack_rows  # noqa: E402

n = int(float(os.environ.get("AB_POINTS", "1e8")))
wl = bench.Workload(os.environ.get("AB_WORKLOAD", bench.HEADLINE), n, 1234, torch.device("cuda", 0), 0)
key = wl.rows_key
base = [g.clone() for g in wl.grads]


def timed(unpacked, k=6):
    for i in range(2):
        wl.launch(i, unpacked=unpacked)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for i, (a, b) in enumerate(ev):
        a.record()
        wl.launch(i, unpacked=unpacked)
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / k


def plastic_fraction():
    h = wl.law._handle(0)
    wl.launch(0)
    torch.cuda.synchronize()
    return h.last_stats().n_plastic / n


for ever_all in (False, True):
    if ever_all:  # every point has yielded once: all rows non-zero, both layouts rebuilt from the same plain array
        hc, ht, mask = wl.plain_twin()
        rows = hc[key]
        rows.view(-1, 6)[:, 0].add_(1e-9)
        ht[key].copy_(rows)
        mask.zero_()
        wl.hist_c[key], wl.ever_c = pack_rows(rows.clone())
        wl.ever_t = wl.ever_c.clone()
        wl.hist_t[key].copy_(wl.hist_c[key])
        wl.hmask.zero_()
    for scale in [float(x) for x in os.environ.get("AB_SCALES", "1.0,0.35,0.2").split(",")]:
        for i in range(2):
            wl.grads[i].copy_(base[i] * scale)
        res = {False: [], True: []}
        for rnd in range(4):
            for unpacked in (False, True):
                res[unpacked].append(timed(unpacked))
        med = {u: sorted(v)[len(v) // 2] for u, v in res.items()}
        ever = float(torch.count_nonzero(wl.plain_twin()[0][key].view(-1, 6).abs().sum(1)) / n) if not ever_all else 1.0
        print(f"EVER {ever:.2f}  plastic now {plastic_fraction():.3f}: packed {med[False]:.3f} ms   reference layout {med[True]:.3f} ms   "
              f"packed / reference {med[False] / med[True]:.3f}", flush=True)
