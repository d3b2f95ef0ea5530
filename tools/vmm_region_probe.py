#!/usr/bin/env python3
"""Is the device's physical memory uniform?  A pool of 2 MiB handles created in one burst is scanned with a window of
W handles: per window position the bandwidth of a pure write (fill), a pure read (sum) and a copy from a fixed
hipMalloc source, next to the time of the evaluate kernel when ONLY its tangent lives in a 14 GB window there
(tools/vmm_offset_sweep.py).       python tools/vmm_region_probe.py [pool GiB] [window MiB]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
pool_gib = float(sys.argv[1]) if len(sys.argv) > 1 else 64
W = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) * (1 << 20) // G
dev = torch.device("cuda", 0)
torch.cuda.init()
src = torch.ones(W * G // 8, dtype=torch.float64, device=dev)  # fixed hipMalloc source of the copies
prop, acc = Prop(1, 0, Loc(1, 0), None, 0, 0, 0), Access(Loc(1, 0), 3)
pool = []
for _ in range(int(pool_gib * (1 << 30)) // G):
    h = C.c_void_p()
    chk(hip.hipMemCreate(C.byref(h), C.c_size_t(G), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
    pool.append(h)


def bw(fn, nbytes, reps=5):
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return round(nbytes / min(a.elapsed_time(b) for a, b in ev) / 1e6, 1)  # GB/s


for k in range(0, len(pool) - W + 1, W):
    va = C.c_void_p()
    chk(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(W * G), C.c_size_t(G), None, C.c_ulonglong(0)), "reserve")
    for i in range(W):
        p = C.c_void_p(va.value + i * G)
        chk(hip.hipMemMap(p, C.c_size_t(G), C.c_size_t(0), pool[k + i], C.c_ulonglong(0)), "map")
        chk(hip.hipMemSetAccess(p, C.c_size_t(G), C.byref(acc), C.c_size_t(1)), "access")
    w = tensor_from_pointer(va.value, W * G // 8, dev)
    rec = {"window_GiB_offset": round(k * G / (1 << 30), 2), "fill_GBs": bw(lambda: w.fill_(2.0), W * G),
           "sum_GBs": bw(lambda: w.sum(), W * G), "copy_from_hipmalloc_GBs": bw(lambda: w.copy_(src), 2 * W * G)}
    print(json.dumps(rec), flush=True)
    del w
    torch.cuda.synchronize()
    for i in range(W):
        chk(hip.hipMemUnmap(C.c_void_p(va.value + i * G), C.c_size_t(G)), "unmap")
