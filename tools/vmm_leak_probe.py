#!/usr/bin/env python3
"""Is the physical memory of a VMM working set returned when it is freed -- also under rocprofv3?"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenics_constitutive_amd import _capi
from fenics_constitutive_amd.placement import VmmArraySet
torch.cuda.init()
ctx = _capi.get_context(0)
for i in range(4):
    free0, _ = torch.cuda.mem_get_info()
    s = VmmArraySet(ctx, {"a": 4_000_000_000, "b": 1_000_000_000}, interleaved=True)
    t = s["a"]; t.fill_(1.0); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    del t
    try:
        s.free()
        err = None
    except Exception as e:
        err = str(e)[:200]
    free2, _ = torch.cuda.mem_get_info()
    print(json.dumps({"round": i, "free_before_GB": round(free0 / 1e9, 1), "free_with_set_GB": round(free1 / 1e9, 1),
                      "free_after_free_GB": round(free2 / 1e9, 1), "error": err}), flush=True)
