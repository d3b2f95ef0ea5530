#!/usr/bin/env python3
"""When all tangent candidates of a process are slow, is the placement of the READ arrays the bad partner?
Four tangent candidates are timed against the original read arrays (gradient, committed stress and history)
and against clones of them in fresh memory.  One JSON line per process; run it several times."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
kind, scale, _, _ = bench.WORKLOADS["von_mises_mixed"]
law, _ = bench.make_law(kind)
grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1234, dev)
cands = [torch.empty(36 * n, dtype=torch.float64, device=dev)]
gw = grad_array()
law.evaluate(0, 2.0, gw, s0, cands[0], h0)
del gw
g = grad_array()
s1 = torch.empty_like(s0)
h1 = {k: v.clone() for k, v in h0.items()}
mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
cands += [torch.empty(36 * n, dtype=torch.float64, device=dev) for _ in range(3)]


def ms(t, gg, ss, hh):
    fn = lambda: law.evaluate_from(0, 2.0, gg, ss, s1, t, hh, h1, history_mask=mask)  # noqa: E731
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for x, y in ev:
        x.record()
        fn()
        y.record()
    torch.cuda.synchronize()
    return round(min(x.elapsed_time(y) for x, y in ev), 3)


a = [ms(t, g, s0, h0) for t in cands]
g2, s02, h02 = g.clone(), s0.clone(), {k: v.clone() for k, v in h0.items()}
b = [ms(t, g2, s02, h02) for t in cands]
a2 = [ms(t, g, s0, h0) for t in cands]
print(json.dumps({"original_reads": a, "cloned_reads": b, "original_again": a2, "best": [min(a), min(b)]}), flush=True)
