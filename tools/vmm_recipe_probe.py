#!/usr/bin/env python3
"""Which VMM recipe?  bench.py's headline step (VonMises3D mixed, sparse trial history, two alternating Newton
iterates) timed on working sets built in different ways (fcamd_device_alloc_set), next to hipMalloc
candidates of the tangent.  A recipe is a list of VMM sets; a set is (array names, granule MiB, interleaved);
arrays that are in no set stay in torch (hipMalloc) memory.

    python tools/vmm_recipe_probe.py [n] [torch candidates]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fenics_constitutive_amd.placement import VmmArraySet  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
k_torch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = bench.Workload("von_mises_mixed", n, 1234, dev, 0)
ctx = wl.law._handle(0).ctx
base = {"tangent": wl.tangent, "stress_c": wl.stress_c, "stress_t": wl.stress_t, "grad0": wl.grads[0], "grad1": wl.grads[1],
        "eps_c": wl.hist_c["eps_n"], "eps_t": wl.hist_t["eps_n"], "alpha_c": wl.hist_c["alpha"], "alpha_t": wl.hist_t["alpha"]}
mask0 = wl.hmask.clone()


def time_on(arrays, label, extra=None):
    wl.tangent, wl.stress_c, wl.stress_t = arrays["tangent"], arrays["stress_c"], arrays["stress_t"]
    wl.grads = [arrays["grad0"], arrays["grad1"]]
    wl.hist_c = {"eps_n": arrays["eps_c"], "alpha": arrays["alpha_c"]}
    wl.hist_t = {"eps_n": arrays["eps_t"], "alpha": arrays["alpha_t"]}
    wl.hmask.copy_(mask0)
    wl.warmup(2)
    ms = wl.timed_events(8)
    rec = {"recipe": label, "kernel_ms_avg": round(sum(ms) / len(ms), 4), "kernel_ms_min": round(min(ms), 4)}
    rec.update(extra or {})
    print(json.dumps(rec), flush=True)
    return rec["kernel_ms_avg"]


res = {}
cands = [base["tangent"]] + [torch.empty(36 * n, dtype=torch.float64, device=dev) for _ in range(k_torch - 1)]
for i, t in enumerate(cands):
    res[f"hipmalloc/{i}"] = time_on({**base, "tangent": t}, f"hipmalloc/{i}")
del cands, t
base["tangent"] = torch.empty(36 * n, dtype=torch.float64, device=dev)
torch.cuda.empty_cache()

W = ["tangent", "stress_t", "eps_t", "alpha_t"]
C = ["stress_c", "eps_c", "alpha_c"]
G = ["grad0", "grad1"]
RECIPES = {
    "all/int2M": [(W + C + G, 2, True)],
    "state/int2M": [(W + C, 2, True)],
    "written/int2M": [(W, 2, True)],
    "all/int16M": [(W + C + G, 16, True)],
    "all/int64M": [(W + C + G, 64, True)],
    "all/seq2M": [(W + C + G, 2, False)],
    "T|rest/int2M": [(["tangent"], 2, False), (["stress_t", "eps_t", "alpha_t"] + C + G, 2, True)],
    "written|read/int2M": [(W, 2, True), (C + G, 2, True)],
    "all/int2M#2": [(W + C + G, 2, True)],
}
for label, sets in RECIPES.items():
    t_ = time.perf_counter()
    arrays, owned = dict(base), []
    for names, gran, inter in sets:
        s = VmmArraySet(ctx, {k: base[k].numel() for k in names}, granule=gran << 20, interleaved=inter)
        owned.append(s)
        for k in names:
            v = s[k]
            if k != "tangent":
                v.copy_(base[k])
            arrays[k] = v
    torch.cuda.synchronize()
    res[label] = time_on(arrays, label, {"build_s": round(time.perf_counter() - t_, 2)})
    del arrays, v, s, owned
    wl.tangent = wl.stress_c = wl.stress_t = wl.grads = wl.hist_c = wl.hist_t = None
hm = [v for k, v in res.items() if k.startswith("hipmalloc/")]
print(json.dumps({"summary": {"n": n, "hipmalloc_ms": hm, "recipes": {k: v for k, v in res.items() if not k.startswith("hipmalloc/")}}}), flush=True)
