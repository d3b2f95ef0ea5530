#!/usr/bin/env python3
"""Throughput of the secondary kernels (low-dimensional constraints, strain_from_grad_u, row maps,
component maps) -- GB/s of algorithmic traffic, for DESIGN.md."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenics_constitutive_amd as fc  # noqa: E402
from fenics_constitutive_amd import _capi  # noqa: E402
from fenics_constitutive_amd.maps import DeviceSubSpaceMap  # noqa: E402

dev = torch.device("cuda", 0)
C = fc.StressStrainConstraint


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms)


def report(name, n, bytes_per_pt, ms):
    print(json.dumps({"kernel": name, "n": n, "ms": round(ms, 3), "Gpts_s": round(n / ms / 1e6, 2),
                      "GB_s": round(n * bytes_per_pt / ms / 1e6, 1)}), flush=True)


def z(m):
    return torch.zeros(m, dtype=torch.float64, device=dev)


n = 200_000_000
SLS = {"E0": 42.0, "E1": 10.0, "tau": 10.0, "nu": 0.2}
for cname, gd2, sd in (("PLANE_STRAIN", 4, 4), ("PLANE_STRESS", 4, 4), ("UNIAXIAL_STRAIN", 1, 1)):
    c = C[cname]
    g = torch.randn(gd2 * n, dtype=torch.float64, device=dev) * 1e-3
    s, t = z(sd * n), z(sd * sd * n)
    le = fc.LinearElasticityModel({"E": 42.0, "nu": 0.3}, c)
    report(f"LE {cname}", n, 8 * (gd2 + 2 * sd + sd * sd), timeit(lambda: le.evaluate(0, 1, g, s, t, None)))
    h = {"strain_visco": z(sd * n), "strain": z(sd * n)}
    for cls in (fc.SpringMaxwellModel, fc.SpringKelvinModel):
        m = cls(SLS, c)
        report(f"{cls.__name__} {cname}", n, 8 * (gd2 + 6 * sd + sd * sd), timeit(lambda: m.evaluate(0, 2.0, g, s, t, h)))
    del g, s, t, h
n = 100_000_000
g = torch.randn(9 * n, dtype=torch.float64, device=dev)
report("strain_from_grad_u FULL", n, 8 * 15, timeit(lambda: fc.strain_from_grad_u(g, C.FULL)))
del g
torch.cuda.empty_cache()
# submesh maps: half of the parent rows, sorted (dolfinx cell order) and shuffled
n_parent, n_sub = 60_000_000, 30_000_000
for order in ("sorted", "random"):
    idx = torch.randperm(n_parent, device=dev)[:n_sub]
    if order == "sorted":
        idx = idx.sort().values
    m = DeviceSubSpaceMap(idx.to(torch.int32), torch.arange(n_sub, dtype=torch.int32, device=dev), device=dev)
    for size in (6, 36):
        parent, sub = z(size * n_parent), z(size * n_sub)
        report(f"map_to_sub rows of {size} ({order})", n_sub, 16 * size + 8, timeit(lambda: m.map_to_sub(parent, sub, size)))
        report(f"map_to_parent rows of {size} ({order})", n_sub, 16 * size + 8, timeit(lambda: m.map_to_parent(sub, parent, size)))
        del parent, sub
# wrappers' component maps
n = 100_000_000
ctx = _capi.get_context(0)
import ctypes as Cc

g2, g3 = z(4 * n), z(9 * n)
report("convert GRAD_2D_TO_3D", n, 64, timeit(lambda: _capi.check(ctx._lib.fcamd_convert_device(ctx.handle, _capi.GRAD_2D_TO_3D, n, Cc.c_void_p(g2.data_ptr()), Cc.c_void_p(g3.data_ptr())))))
del g2, g3
t3, t2 = z(36 * n), z(16 * n)
report("convert TANGENT_3D_TO_2D", n, 256, timeit(lambda: _capi.check(ctx._lib.fcamd_convert_device(ctx.handle, _capi.TANGENT_3D_TO_2D, n, Cc.c_void_p(t3.data_ptr()), Cc.c_void_p(t2.data_ptr())))))
