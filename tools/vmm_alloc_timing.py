#!/usr/bin/env python3
"""How long does fcamd_device_alloc_set take per physical handle?  (sizes x granules; one JSON line each)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenics_constitutive_amd import _capi  # noqa: E402

torch.cuda.init()
ctx = _capi.get_context(0)
for gib, granule_mib in ((0.25, 2), (1, 2), (4, 2), (4, 64), (16, 64), (16, 1024), (16, 2), (64, 1024)):
    nbytes = int(gib * (1 << 30))
    t0 = time.perf_counter()
    ptrs = ctx.alloc_set([nbytes], granule_mib << 20, interleaved=False)
    t1 = time.perf_counter()
    x = __import__("fenics_constitutive_amd.placement", fromlist=["tensor_from_pointer"]).tensor_from_pointer(ptrs[0], nbytes // 8, "cuda:0")
    x.fill_(1.0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    del x
    ctx.free(ptrs[0])
    t3 = time.perf_counter()
    print(json.dumps({"GiB": gib, "granule_MiB": granule_mib, "handles": nbytes // (granule_mib << 20), "alloc_s": round(t1 - t0, 3),
                      "first_fill_s": round(t2 - t1, 3), "free_s": round(t3 - t2, 3)}), flush=True)
