#!/usr/bin/env python3
"""How much headroom is there in placement?  Greedy, array by array (largest stream first): K candidate regions of a pool
of 2 MiB handles per array, the real kernel timed on each, the best kept, the others fixed.  Start: everything in
hipMalloc memory.       python tools/vmm_greedy_probe.py [n] [K] [--sparse]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fenics_constitutive_amd.placement import tensor_from_pointer  # noqa: E402
from vmm_raw import G, create_handles, map_handles, unmap  # noqa: E402

args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(float(args_[0])) if args_ else 50_000_000
K = int(args_[1]) if len(args_) > 1 else 6
dev = torch.device("cuda", 0)
wl = bench.Workload("von_mises_mixed", n, 1234, dev, 0, history="sparse" if "--sparse" in sys.argv else "full")


def new_region(numel):
    """a fresh array of `numel` doubles on its own burst of 2 MiB handles"""
    hs = create_handles(-(-numel * 8 // G))
    va = map_handles(hs)
    return tensor_from_pointer(va, numel, dev), (va, hs)


def drop(region):
    va, hs = region
    unmap(va, len(hs), release=hs)


def get(name):
    return {"tangent": wl.tangent, "grad0": wl.grads[0], "grad1": wl.grads[1], "stress_t": wl.stress_t, "stress_c": wl.stress_c,
            "eps_t": wl.hist_t["eps_n"], "eps_c": wl.hist_c["eps_n"]}[name]


def put(name, t):
    if name == "tangent":
        wl.tangent = t
    elif name in ("grad0", "grad1"):
        wl.grads[int(name[-1])] = t
    elif name == "stress_t":
        wl.stress_t = t
    elif name == "stress_c":
        wl.stress_c = t
    elif name == "eps_t":
        wl.hist_t["eps_n"] = t
    else:
        wl.hist_c["eps_n"] = t


def kernel_ms():
    wl.warmup(2)
    ms = wl.timed_events(6)
    return round(sum(ms) / len(ms), 4)


base = kernel_ms()
print(json.dumps({"step": "start (hipMalloc arrays)", "kernel_ms": base}), flush=True)
cur = base
for name in ("tangent", "grad0", "grad1", "stress_t", "eps_t", "stress_c", "eps_c"):
    original = get(name)
    cands, times = [], []
    for k in range(K):
        t, region = new_region(original.numel())
        t.copy_(original)
        put(name, t)
        cands.append((t, region))
        times.append(kernel_ms())
    best = min(range(K), key=lambda k: times[k])
    if times[best] < cur:
        put(name, cands[best][0])
        cur = times[best]
        keep = best
    else:
        put(name, original)
        keep = None
    for k, (t, region) in enumerate(cands):
        if k != keep:
            del t
    cands_regions = [c[1] for k, c in enumerate(cands) if k != keep]
    cands = None
    for region in cands_regions:
        drop(region)
    print(json.dumps({"step": name, "candidate_ms": times, "kept": keep, "kernel_ms": cur}), flush=True)
print(json.dumps({"summary": {"n": n, "start_ms": base, "final_ms": cur, "gain_pct": round(100 * (base / cur - 1), 2)}}), flush=True)
