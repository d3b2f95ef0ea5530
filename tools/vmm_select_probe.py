#!/usr/bin/env python3
"""Can a cheap pairwise test pick good memory for the tangent?  A pool of 2 MiB handles is scanned with a 1 GiB window;
per window: bandwidth of a copy FROM the workload's own gradient array into the window.  Then the tangent is built
(hipMemMap) from the best-ranked windows, from the worst-ranked ones and from the first ones, and the evaluate kernel is
timed on each.      python tools/vmm_select_probe.py [n] [pool GiB]"""
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
W = 512  # handles per window = 1 GiB
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
pool_gib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
wl = bench.Workload("von_mises_mixed", n, 1234, dev, 0, history="full")
prop, acc = Prop(1, 0, Loc(1, 0), None, 0, 0, 0), Access(Loc(1, 0), 3)
pool = []
for _ in range(pool_gib * W):
    h = C.c_void_p()
    chk(hip.hipMemCreate(C.byref(h), C.c_size_t(G), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
    pool.append(h)


def map_handles(handles):
    va = C.c_void_p()
    chk(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(len(handles) * G), C.c_size_t(G), None, C.c_ulonglong(0)), "reserve")
    for i, h in enumerate(handles):
        p = C.c_void_p(va.value + i * G)
        chk(hip.hipMemMap(p, C.c_size_t(G), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
        chk(hip.hipMemSetAccess(p, C.c_size_t(G), C.byref(acc), C.c_size_t(1)), "access")
    return va.value


def unmap(va, count):
    torch.cuda.synchronize()
    for i in range(count):
        chk(hip.hipMemUnmap(C.c_void_p(va + i * G), C.c_size_t(G)), "unmap")


def bw(fn, nbytes, reps=4):
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return nbytes / min(a.elapsed_time(b) for a, b in ev) / 1e6


numel_w = W * G // 8
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
