import json, os, sys, time
import torch
sys.path.insert(0, "/root/repo")
import bench
from fenics_constitutive_amd.placement import VmmArraySet
dev = torch.device("cuda", 0)
n = 100_000_000
wl = bench.Workload("von_mises_mixed", n, 1234, dev, 0)
ctx = wl.law._handle(0).ctx
base = {"tangent": wl.tangent, "stress_c": wl.stress_c, "stress_t": wl.stress_t, "grad0": wl.grads[0], "grad1": wl.grads[1],
        "eps_c": wl.hist_c["eps_n"], "eps_t": wl.hist_t["eps_n"], "alpha_c": wl.hist_c["alpha"], "alpha_t": wl.hist_t["alpha"]}
mask0 = wl.hmask.clone()
def time_on(arrays, label):
    wl.tangent, wl.stress_c, wl.stress_t = arrays["tangent"], arrays["stress_c"], arrays["stress_t"]
    wl.grads = [arrays["grad0"], arrays["grad1"]]
    wl.hist_c = {"eps_n": arrays["eps_c"], "alpha": arrays["alpha_c"]}
    wl.hist_t = {"eps_n": arrays["eps_t"], "alpha": arrays["alpha_t"]}
    wl.hmask.copy_(mask0)
    wl.warmup(2)
    ms = wl.timed_events(8)
    print(json.dumps({"case": label, "ms": round(sum(ms)/len(ms), 4)}), flush=True)
cands = [base["tangent"]] + [torch.empty(36 * n, dtype=torch.float64, device=dev) for _ in range(5)]
for i, t in enumerate(cands):
    time_on({**base, "tangent": t}, f"hipmalloc/{i}")
del cands, t
torch.cuda.empty_cache()
names = list(base)
def build(tag):
    s = VmmArraySet(ctx, {k: base[k].numel() for k in names}, interleaved=True)
    arrays = {}
    for k in names:
        v = s[k]
        if k != "tangent": v.copy_(base[k])
        arrays[k] = v
    return s, arrays
A, a = build("A"); time_on(a, "A (first set, after freeing the hipMalloc candidates)")
B, b = build("B"); time_on(b, "B (second set, A alive)")
time_on(a, "A again")
del a; A.free(); del A
C, c = build("C"); time_on(c, "C (after freeing A, B alive)")
del b; B.free(); del B
time_on(c, "C again (B freed)")
D, d = build("D"); time_on(d, "D (C alive)")
del c; C.free(); del C; del d; D.free(); del D
wl.tangent = wl.stress_c = wl.stress_t = wl.grads = wl.hist_c = wl.hist_t = None
E, e = build("E"); time_on(e, "E (everything else freed)")
