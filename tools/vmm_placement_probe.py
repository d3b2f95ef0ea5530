#!/usr/bin/env python3
"""does building the working set through the virtual-memory API
(fcamd_device_alloc_set: hipMemAddressReserve + hipMemCreate / hipMemMap in granule-sized handles, created
array after array or interleaved over the arrays) remove the placement lottery of DESIGN.md 6?

    python tools/vmm_placement_probe.py [n] [torch candidates]

Same kernel, same data, the mask-less committed -> trial VonMises3D step of bench.py's headline workload;
only WHERE the arrays live changes:
    torch/k          tangent = k-th torch.empty candidate (all alive together), everything else torch
    vmm_w/seq        the four written arrays (tangent, trial stress, trial eps_n, trial alpha) in one VMM set,
                     2 MiB handles created array after array; read arrays torch
    vmm_w/int        the same, handles interleaved over the four arrays
    vmm_all/int[/G]  all eight arrays of the step in one VMM set, interleaved, granule G (2 MiB default)
Every placement: 1 warm + 4 timed launches (tools/summarize_placement_pmc.py relies on 5 dispatches per
placement, after the one in-place warm step of the set-up).  One JSON line per placement."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fenics_constitutive_amd.placement import VmmArraySet  # noqa: E402

dev = torch.device("cuda", 0)
if "--selftest" in sys.argv:
    # every array of a VMM set must be backed by its own physical memory: distinct patterns, verified after
    # ALL arrays have been written (an interleaved set once aliased, see fcamd_memory.cpp)
    from fenics_constitutive_amd import _capi
    from fenics_constitutive_amd.placement import tensor_from_pointer

    torch.cuda.init()
    ctx_ = _capi.get_context(0)
    ok = True
    for n_ in (100_000, 3_000_000, 20_000_000):
        sizes = [8 * 36 * n_, 8 * 6 * n_, 8 * 6 * n_, 8 * n_, 8 * 9 * n_]
        for granule in (0, 64 << 20):
            for inter in (False, True):
                ptrs = ctx_.alloc_set(sizes, granule, interleaved=inter)
                ts = [tensor_from_pointer(p, s // 8, dev) for p, s in zip(ptrs, sizes)]
                for k, t in enumerate(ts):
                    t.copy_(torch.arange(t.numel(), dtype=torch.float64, device=dev) + k * 1e10)
                torch.cuda.synchronize()
                bad = [k for k, t in enumerate(ts) if not torch.equal(t, torch.arange(t.numel(), dtype=torch.float64, device=dev) + k * 1e10)]
                print(json.dumps({"selftest_n": n_, "granule_MiB": (granule or (2 << 20)) >> 20, "interleaved": inter, "aliased_arrays": bad}), flush=True)
                ok = ok and not bad
                del ts, t
                for p in ptrs:
                    ctx_.free(p)
    sys.exit(0 if ok else 1)
args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(float(args_[0])) if args_ else 50_000_000
k_torch = int(args_[1]) if len(args_) > 1 else 6
reps = int(args_[2]) if len(args_) > 2 else 3
SPARSE = "--sparse" in sys.argv  # the headline step of bench.py (sparse trial history) instead of the mask-less one
kind, scale, _, _ = bench.WORKLOADS["von_mises_mixed"]
law, _ = bench.make_law(kind)
grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1234, dev)
t0 = torch.empty(36 * n, dtype=torch.float64, device=dev)
gw = grad_array()
law.evaluate(0, 2.0, gw, s0, t0, h0)  # the one in-place warm step: committed state "from a previous step"
del gw
g = grad_array()
s1, e1, a1 = torch.empty_like(s0), torch.empty_like(h0["eps_n"]), torch.empty_like(h0["alpha"])
ctx = law._handle(0).ctx
mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev) if SPARSE else None


def run(label, grad, sc, ec, ac, st, tan, et, at, extra=None):
    if SPARSE:  # contract of the sparse protocol: trial == committed where the mask is clear
        et.copy_(ec), at.copy_(ac), mask.zero_()

    def step():
        law.evaluate_from(0, 2.0, grad, sc, st, tan, {"eps_n": ec, "alpha": ac}, {"eps_n": et, "alpha": at}, history_mask=mask)

    step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
    for x, y in ev:
        x.record()
        step()
        y.record()
    torch.cuda.synchronize()
    ms = [x.elapsed_time(y) for x, y in ev]
    rec = {"placement": label, "kernel_ms_min": round(min(ms), 4), "kernel_ms_avg": round(sum(ms) / 4, 4), "tangent_ptr": hex(tan.data_ptr())}
    rec.update(extra or {})
    print(json.dumps(rec), flush=True)
    return min(ms)


results = {}
cands = [t0] + [torch.empty(36 * n, dtype=torch.float64, device=dev) for _ in range(k_torch - 1)]
for i, t in enumerate(cands):
    results[f"torch/{i}"] = run(f"torch/{i}", g, s0, h0["eps_n"], h0["alpha"], s1, t, e1, a1)
ref_s, ref_t = s1.clone(), cands[-1].clone()
del cands, t0, t
torch.cuda.empty_cache()

written = {"tangent": 36 * n, "stress_t": 6 * n, "eps_t": 6 * n, "alpha_t": n}
committed = {"stress_c": 6 * n, "eps_c": 6 * n, "alpha_c": n}
read = {"grad": 9 * n, **committed}


def vmm_case(label, numels, granule, interleaved):
    import time

    t_ = time.perf_counter()
    aset = VmmArraySet(ctx, numels, granule=granule, interleaved=interleaved)
    alloc_s = time.perf_counter() - t_
    try:
        gr = sc = ec = ac = None
        if "grad" in numels:
            gr = aset["grad"]
            gr.copy_(g)
        if "stress_c" in numels:
            sc, ec, ac = aset["stress_c"], aset["eps_c"], aset["alpha_c"]
            sc.copy_(s0), ec.copy_(h0["eps_n"]), ac.copy_(h0["alpha"])
        st, tan, et, at = aset["stress_t"], aset["tangent"], aset["eps_t"], aset["alpha_t"]
        results[label] = run(label, g if gr is None else gr, s0 if sc is None else sc, h0["eps_n"] if ec is None else ec,
                             h0["alpha"] if ac is None else ac, st, tan, et, at,
                             {"alloc_s": round(alloc_s, 2), "granule_MiB": (granule or (2 << 20)) >> 20})
        assert torch.equal(st, ref_s) and torch.equal(tan, ref_t), "VMM placement changed the results"
    finally:
        del gr, sc, ec, ac, st, tan, et, at
        aset.free()


# fresh allocations of the same recipe: is the time a property of the recipe?
for rep in range(reps):
    vmm_case(f"vmm_w/int#{rep}", written, 0, True)       # the four written arrays; everything that is read: torch
for rep in range(reps):
    vmm_case(f"vmm_state/int#{rep}", {**written, **committed}, 0, True)  # what a resident state owns: both copies + tangent
for rep in range(reps):
    vmm_case(f"vmm_all/int#{rep}", {**written, **read}, 0, True)
vmm_case("vmm_w/seq", written, 0, False)
vmm_case("vmm_all/seq", {**written, **read}, 0, False)
vmm_case("vmm_w/int/64M", written, 64 << 20, True)
vmm_case("vmm_w/int/1G", written, 1 << 30, True)

tv = [v for k, v in results.items() if k.startswith("torch/")]


def grp(prefix):
    v = [x for k, x in results.items() if k.startswith(prefix)]
    return {"ms": [round(x, 4) for x in v], "worst_vs_torch_best_pct": round(100 * (max(v) / min(tv) - 1), 2)}


print(json.dumps({"summary": {"n": n, "step": "sparse" if SPARSE else "maskless", "torch_ms": [round(x, 4) for x in tv],
                              "torch_spread_pct": round(100 * (max(tv) / min(tv) - 1), 2),
                              "vmm_w_int": grp("vmm_w/int#"), "vmm_state_int": grp("vmm_state/int#"), "vmm_all_int": grp("vmm_all/int#"),
                              "others": {k: round(v, 4) for k, v in results.items() if k.count("/") == 2 or k.endswith("/seq")}}}), flush=True)
