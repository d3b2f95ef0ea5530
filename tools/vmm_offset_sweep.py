#!/usr/bin/env python3
"""Does the PHYSICAL offset of the tangent relative to the other arrays decide the kernel time?

The tangent is mapped (hipMemMap, straight through ctypes on the HIP runtime) onto a window of a pool of 2 MiB
physical handles created in one burst; shifting the window by k handles moves the array by k x 2 MiB in
whatever physical order the driver handed the pool out.  Everything else (hipMalloc arrays, data, kernel)
stays.  One JSON line per shift.      python tools/vmm_offset_sweep.py [n] [max_shift]
Every shift gets a fresh address range (a re-used range serves stale data on this stack)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fenics_constitutive_amd.placement import tensor_from_pointer  # noqa: E402

hip = C.CDLL("libamdhip64.so.7")


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("handle_type", C.c_int), ("location", Loc), ("win32", C.c_void_p),
                ("compression", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def chk(e, what):
    if e != 0:
        raise RuntimeError(f"{what} -> hip error {e}")


G = 2 << 20
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
max_shift = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
wl = bench.Workload("von_mises_mixed", n, 1234, dev, 0, history="full")
T = -(-36 * n * 8 // G)
prop = Prop(1, 0, Loc(1, 0), None, 0, 0, 0)  # pinned, device 0
acc = Access(Loc(1, 0), 3)
pool = []
for _ in range(T + max_shift):
    h = C.c_void_p()
    chk(hip.hipMemCreate(C.byref(h), C.c_size_t(G), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
    pool.append(h)
print(json.dumps({"pool_handles": len(pool), "tangent_handles": T, "n": n}), flush=True)


def map_window(k, order="forward"):
    """tangent granule i <- pool handle k + perm(i): forward, reversed, or a fixed random permutation of the window"""
    va = C.c_void_p()
    chk(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(T * G), C.c_size_t(G), None, C.c_ulonglong(0)), "reserve")
    if order == "forward":
        perm = range(T)
    elif order == "reversed":
        perm = range(T - 1, -1, -1)
    else:
        import random

        perm = list(range(T))
        random.Random(7).shuffle(perm)
    for i, j in zip(range(T), perm):
        p = C.c_void_p(va.value + i * G)
        chk(hip.hipMemMap(p, C.c_size_t(G), C.c_size_t(0), pool[j + k], C.c_ulonglong(0)), "map")
        chk(hip.hipMemSetAccess(p, C.c_size_t(G), C.byref(acc), C.c_size_t(1)), "access")
    return va.value


def unmap_window(va):
    torch.cuda.synchronize()
    for i in range(T):
        chk(hip.hipMemUnmap(C.c_void_p(va + i * G), C.c_size_t(G)), "unmap")


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
