#!/usr/bin/env python3
"""Long run on fixed buffers: does the kernel time switch modes over time (clock/power state)?"""
import os, sys, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
n = 100_000_000
kind, scale, _, _ = bench.WORKLOADS["von_mises_mixed"]
law, _ = bench.make_law(kind)
grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1, dev)
g = grad_array()
t = torch.empty(36 * n, dtype=torch.float64, device=dev)
s1 = torch.empty_like(s0)
h1 = {k: torch.empty_like(v) for k, v in h0.items()}
for block in range(16):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(); law.evaluate_from(0, 2.0, g, s0, s1, t, h0, h1); b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    clk = ""
    if block % 4 == 3:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            clk = " | " + " ".join(l.split(":")[-1].strip() for l in out.splitlines() if "clock level" in l or "Power" in l)
        except Exception as e:
            clk = f" | smi failed {e}"
    print(f"block {block:2d}: min {min(ms):.3f} med {sorted(ms)[15]:.3f} max {max(ms):.3f}{clk}", flush=True)
