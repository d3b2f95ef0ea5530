#!/usr/bin/env python3
"""Is the device's physical memory uniform?  A pool of 2 MiB handles created in one burst is scanned with a window of
W handles: per window position the bandwidth of a pure write (fill), a pure read (sum) and a copy from a fixed
hipMalloc source, next to the time of the evaluate kernel when ONLY its tangent lives in a 14 GB window there
(tools/vmm_offset_sweep.py).       python tools/vmm_region_probe.py [pool GiB] [window MiB]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenics_constitutive_amd.placement import tensor_from_pointer  # noqa: E402
from vmm_raw import G, bandwidth_GBs, create_handles, map_handles, unmap  # noqa: E402

pool_gib = float(sys.argv[1]) if len(sys.argv) > 1 else 64
W = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) * (1 << 20) // G
dev = torch.device("cuda", 0)
torch.cuda.init()
src = torch.ones(W * G // 8, dtype=torch.float64, device=dev)  # fixed hipMalloc source of the copies
pool = create_handles(int(pool_gib * (1 << 30)) // G)


def bw(fn, nbytes):
    return round(bandwidth_GBs(fn, nbytes, reps=5), 1)


for k in range(0, len(pool) - W + 1, W):
    va = map_handles(pool[k:k + W])
    w = tensor_from_pointer(va, W * G // 8, dev)
    rec = {"window_GiB_offset": round(k * G / (1 << 30), 2), "fill_GBs": bw(lambda: w.fill_(2.0), W * G),
           "sum_GBs": bw(lambda: w.sum(), W * G), "copy_from_hipmalloc_GBs": bw(lambda: w.copy_(src), 2 * W * G)}
    print(json.dumps(rec), flush=True)
    del w
    unmap(va, W)
