#!/usr/bin/env python3
"""Memory-side counters of the headline kernel on a GOOD and on a BAD hipMalloc draw of the tangent array, in one process.

    rocprofv3 --pmc <COUNTER> --output-format csv -d <dir> -- python3 tools/placement_counters.py <json-out>

Builds bench.py's headline workload (VonMises3D mixed, 1e8 points, packed history), times the real launch on K
candidate allocations of the tangent (events, min of 3), then issues 4 launches on the fastest and 4 on the slowest candidate
and writes which evaluate dispatches those were (the profiler's CSV is joined on the dispatch order by
tools/summarize_placement_counters.py).  Without the profiler it just prints the candidate times."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

n = int(float(os.environ.get("PC_POINTS", "1e8")))
K = int(os.environ.get("PC_CANDIDATES", "6"))
dev = torch.device("cuda", 0)
wl = bench.Workload(bench.HEADLINE, n, 1234, dev, 0)
log = list(wl.launch_log)  # evaluate launches so far: the warm in-place step
count = sum(k for _, k in log)
cands = [wl.tangent] + [torch.empty_like(wl.tangent) for _ in range(K - 1)]
ms = []
for t in cands:
    wl.launch(0, tangent=t, sparse_tangent=False)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for a, b in ev:
        a.record()
        wl.launch(0, tangent=t, sparse_tangent=False)
        b.record()
    torch.cuda.synchronize()
    ms.append(min(a.elapsed_time(b) for a, b in ev))
    count += 4
best, worst = ms.index(min(ms)), ms.index(max(ms))
marks = {}
for name, k in (("best", best), ("worst", worst)):
    marks[name] = list(range(count, count + 4))
    for _ in range(4):
        wl.launch(0, tangent=cands[k], sparse_tangent=False)
    count += 4
torch.cuda.synchronize()
out = {"n": n, "candidate_ms": [round(x, 4) for x in ms], "best": best, "worst": worst, "evaluate_dispatch_index": marks}
print(json.dumps(out), flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f)
