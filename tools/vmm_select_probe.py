#!/usr/bin/env python3
"""Can a cheap pairwise test pick good memory for the tangent?  A pool of 2 MiB handles is scanned with a 1 GiB window;
per window: bandwidth of a copy FROM the workload's own gradient array into the window.  Then the tangent is built
(hipMemMap) from the best-ranked windows, from the worst-ranked ones and from the first ones, and the evaluate kernel is
timed on each.      python tools/vmm_select_probe.py [n] [pool GiB]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fenics_constitutive_amd.placement import tensor_from_pointer  # noqa: E402
from vmm_raw import G, bandwidth_GBs as bw, create_handles, map_handles, unmap  # noqa: E402

W = 512  # handles per window = 1 GiB
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
pool_gib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
wl = bench.Workload("von_mises_mixed", n, 1234, dev, 0, history="full")
pool = create_handles(pool_gib * W)
numel_w = W * G // 8
assert 9 * n >= 3 * numel_w, "n too small: the gradient array must hold three 1 GiB source windows (n >= 4.5e7)"
sources = {"grad": wl.grads[0][:numel_w], "stress_c": wl.stress_c[:numel_w], "grad_mid": wl.grads[0][2 * numel_w: 3 * numel_w]}
score = []
for k in range(pool_gib):
    va = map_handles(pool[k * W:(k + 1) * W])
    w = tensor_from_pointer(va, numel_w, dev)
    rec = {"window": k}
    for name, src in sources.items():
        rec["copy_from_" + name] = round(bw(lambda: w.copy_(src), 2 * W * G), 0)
    score.append(rec)
    del w
    unmap(va, W)
print(json.dumps({"windows": score}), flush=True)

T = -(-36 * n * 8 // G)
nw = -(-T // W)
ms0 = wl.timed_events(5)
print(json.dumps({"tangent": "hipmalloc", "kernel_ms": round(min(ms0[1:]), 4)}), flush=True)
for key in ("copy_from_grad", "copy_from_stress_c", "copy_from_grad_mid"):
    ranked = sorted(score, key=lambda r: -r[key])
    for label, chosen in (("best", ranked[:nw]), ("worst", ranked[-nw:])):
        handles = [h for r in sorted(chosen, key=lambda r: r["window"]) for h in pool[r["window"] * W:(r["window"] + 1) * W]][:T]
        va = map_handles(handles)
        wl.tangent = tensor_from_pointer(va, 36 * n, dev)
        ms = wl.timed_events(5)
        print(json.dumps({"tangent": f"{label} {nw} windows by {key}", "kernel_ms": round(min(ms[1:]), 4),
                          "mean_score": round(sum(r[key] for r in chosen) / len(chosen), 0)}), flush=True)
        wl.tangent = None
        unmap(va, T)
for label, start in (("first windows", 0), ("last windows", pool_gib - nw)):
    handles = pool[start * W:start * W + T]
    va = map_handles(handles)
    wl.tangent = tensor_from_pointer(va, 36 * n, dev)
    ms = wl.timed_events(5)
    print(json.dumps({"tangent": label, "kernel_ms": round(min(ms[1:]), 4)}), flush=True)
    wl.tangent = None
    unmap(va, T)
