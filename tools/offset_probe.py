#!/usr/bin/env python3
"""Does the evaluate time depend on the RELATIVE placement of the concurrently streamed arrays?
One array at a time is slid inside an over-allocated buffer by a byte offset delta (everything else,
including the physical pages, stays put) and the kernel is timed.  (variance study.)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
SLACK = (1 << 30) + (1 << 26)  # bytes
kind, scale, _, _ = bench.WORKLOADS["von_mises_mixed"]
law, _ = bench.make_law(kind)
grad_array, s0, h0 = bench.synth_inputs(kind, scale, n, 1234, dev)
tbuf = torch.empty(36 * n + SLACK // 8, dtype=torch.float64, device=dev)
t = tbuf[: 36 * n]
gw = grad_array()
law.evaluate(0, 2.0, gw, s0, t, h0)
del gw
g0 = grad_array()
gbuf = torch.empty(9 * n + SLACK // 8, dtype=torch.float64, device=dev)
s1buf = torch.empty(6 * n + SLACK // 8, dtype=torch.float64, device=dev)
e1, a1 = torch.empty_like(h0["eps_n"]), torch.empty_like(h0["alpha"])


def timed(g, s1, tt):
    run = lambda: law.evaluate_from(0, 2.0, g, s0, s1, tt, h0, {"eps_n": e1, "alpha": a1})  # noqa: E731
    run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
    for x, y in ev:
        x.record()
        run()
        y.record()
    torch.cuda.synchronize()
    return round(sum(x.elapsed_time(y) for x, y in ev) / len(ev), 3)


deltas = [0, 256, 1024, 4096, 16384, 65536, 3 * 65536, 1 << 18, 1 << 19, 1 << 20, 3 << 19, 1 << 21, 3 << 20, 1 << 22, 5 << 20,
          1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 1 << 29, 1 << 30, (1 << 30) + (1 << 21) + 4096]
for name in ("tangent", "grad", "stress_out"):
    row = {}
    for d in deltas:
        gg = gbuf[d // 8: d // 8 + 9 * n] if name == "grad" else gbuf[: 9 * n]
        if name == "grad" or d == 0:
            gg.copy_(g0)
        tt = tbuf[d // 8: d // 8 + 36 * n] if name == "tangent" else tbuf[: 36 * n]
        ss = s1buf[d // 8: d // 8 + 6 * n] if name == "stress_out" else s1buf[: 6 * n]
        row[str(d)] = timed(gg, ss, tt)
    print(json.dumps({"slid": name, "n": n, "ms_by_delta": row}), flush=True)
