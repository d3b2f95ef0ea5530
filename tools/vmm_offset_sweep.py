#!/usr/bin/env python3
"""Does the PHYSICAL offset of the tangent relative to the other arrays decide the kernel time?

The tangent is mapped (hipMemMap, straight through ctypes on the HIP runtime) onto a window of a pool of 2 MiB
physical handles created in one burst; shifting the window by k handles moves the array by k x 2 MiB in
whatever physical order the driver handed the pool out.  Everything else (hipMalloc arrays, data, kernel)
stays.  One JSON line per shift.      python tools/vmm_offset_sweep.py [n] [max_shift]
Every shift gets a fresh address range (a re-used range serves stale data on this stack)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fenics_constitutive_amd.placement import tensor_from_pointer  # noqa: E402
from vmm_raw import G, create_handles, map_handles, unmap  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
max_shift = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
wl = bench.Workload("von_mises_mixed", n, 1234, dev, 0, history="full")
T = -(-36 * n * 8 // G)
pool = create_handles(T + max_shift)
print(json.dumps({"pool_handles": len(pool), "tangent_handles": T, "n": n}), flush=True)


def map_window(k, order="forward"):
    """tangent page i <- pool handle k + perm(i): forward, reversed, or a fixed random permutation of the window"""
    if order == "forward":
        perm = range(T)
    elif order == "reversed":
        perm = range(T - 1, -1, -1)
    else:
        import random

        perm = list(range(T))
        random.Random(7).shuffle(perm)
    return map_handles([pool[j + k] for j in perm])


def unmap_window(va):
    unmap(va, T)


ms0 = wl.timed_events(5)
print(json.dumps({"shift": "hipmalloc", "kernel_ms": round(min(ms0[1:]), 4)}), flush=True)
shifts = [(k, "forward") for k in (0, 1, 7, 64, 1024, 4096)]
# disjoint windows of the pool (different physical memory), and the same window in other granule orders
shifts += [(k, o) for k in range(0, max_shift + 1, T) for o in ("forward", "reversed", "shuffled")]
for k, order in [s for s in shifts if s[0] <= max_shift]:
    va = map_window(k, order)
    wl.tangent = tensor_from_pointer(va, 36 * n, dev)
    ms = wl.timed_events(5)
    print(json.dumps({"shift": k, "order": order, "kernel_ms": round(min(ms[1:]), 4), "avg": round(sum(ms[1:]) / 4, 4)}), flush=True)
    wl.tangent = None
    unmap_window(va)
# the hipMalloc tangent again, and fresh hipMalloc candidates, at the end
wl.tangent = torch.empty(36 * n, dtype=torch.float64, device=dev)
for c in range(4):
    ms = wl.timed_events(5)
    print(json.dumps({"shift": f"hipmalloc_candidate_{c}", "kernel_ms": round(min(ms[1:]), 4)}), flush=True)
    keep = wl.tangent  # keep the previous candidate alive so that the next one is different memory
    wl.tangent = torch.empty(36 * n, dtype=torch.float64, device=dev)
