#!/usr/bin/env python3
"""How much does the evaluate time move when the SAME data is placed in freshly allocated arrays
inside one process?  (run-to-run variance.)  Candidate k is a clone of candidate 0 made
while the earlier candidates are still alive, so the driver has to hand out different memory."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind, scale, _, _ = bench.WORKLOADS["von_mises_mixed"]
law, _ = bench.make_law(kind)
grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1234, dev)
t = torch.empty(36 * n, dtype=torch.float64, device=dev)
gw = grad_array()
law.evaluate(0, 2.0, gw, s0, t, h0)
del gw
base = {"g": grad_array(), "s0": s0, "s1": torch.empty_like(s0), "t": t, "e0": h0["eps_n"], "a0": h0["alpha"],
        "e1": h0["eps_n"].clone(), "a1": h0["alpha"].clone()}


def probe(a, sparse):
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev) if sparse else None
    if sparse:
        a["e1"].copy_(a["e0"]), a["a1"].copy_(a["a0"])
    run = lambda: law.evaluate_from(0, 2.0, a["g"], a["s0"], a["s1"], a["t"], {"eps_n": a["e0"], "alpha": a["a0"]},  # noqa: E731
                                    {"eps_n": a["e1"], "alpha": a["a1"]}, history_mask=mask)
    for _ in range(2):
        run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for x, y in ev:
        x.record()
        run()
        y.record()
    torch.cuda.synchronize()
    return round(sum(x.elapsed_time(y) for x, y in ev) / len(ev), 3)


def copy_bw(a):
    half = (a["t"].numel() // 2) & ~1
    x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(3):
        x.record()
        a["t"][:half].copy_(a["t"][half:2 * half])
        y.record()
        y.synchronize()
        ms = x.elapsed_time(y)
        best = ms if best is None else min(best, ms)
    return round(2 * 8 * half / (best * 1e-3) / 1e12, 2)


sets = [base]  # every candidate stays alive: n * 570 B * tries must fit the 288 GB
for k in range(tries):
    a = sets[-1]
    print(json.dumps({"candidate": k, "full_ms": probe(a, False), "sparse_ms": probe(a, True), "full_again_ms": probe(a, False),
                      "copy_TBs": copy_bw(a), "tangent_ptr": hex(a["t"].data_ptr())}), flush=True)
    if k + 1 < tries:
        sets.append({k2: v.clone() for k2, v in base.items()})

# which array carries the difference?  substitute one array of the slowest candidate into the fastest
times = [probe(a, False) for a in sets]
fast, slow = sets[times.index(min(times))], sets[times.index(max(times))]
print(json.dumps({"fastest_ms": min(times), "slowest_ms": max(times)}), flush=True)
sub = {}
for name in base:
    mixed = dict(fast)
    mixed[name] = slow[name]
    sub[name] = probe(mixed, False)
print(json.dumps({"fast_set_with_one_array_from_slow_set_ms": sub}), flush=True)
sub = {}
for name in base:
    mixed = dict(slow)
    mixed[name] = fast[name]
    sub[name] = probe(mixed, False)
print(json.dumps({"slow_set_with_one_array_from_fast_set_ms": sub}), flush=True)
