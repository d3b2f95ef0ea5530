#!/usr/bin/env python3
"""Memory-side PMC counters of the evaluate kernels, row by row: what the L2 asks of the memory per launch -- requests by size,
stalls -- for the dense kernels next to the rows that move isolated pieces (reference-layout history rows, sparse tangent rows,
permuted parent rows).
    python tools/row_counters.py [--points N] [--out FILE.md] [ITEM ...]
One `rocprofv3 --pmc COUNTER` pass per counter (no trace domain next to --pmc); every pass runs ALL items in one child process
(`bench.py --pmc-child`, 4 timed launches each, the allocator's arrays as they come) and is sliced with the child's launch log
(benchlib/traffic.py).  Run on the GPU box through gpurun."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib.traffic import _pmc_pass, slice_timed  # noqa: E402

COUNTERS = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum",
            "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum",
            "TCC_TAG_STALL_sum", "TCC_BUSY_sum"]
ITEMS = ["linear_elasticity", "von_mises_mixed", "von_mises_mixed+unpacked", "von_mises_mixed+in_place", "resident_sparse_tangent",
         "indexed_scattered_cells", "indexed_permuted"]

ap = argparse.ArgumentParser()
ap.add_argument("items", nargs="*", default=ITEMS)
ap.add_argument("--points", type=int, default=100_000_000)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "row_counters.md"))
ap.add_argument("--counters", default=",".join(COUNTERS))
a = ap.parse_args()
table = {}
for c in a.counters.split(","):
    res = _pmc_pass(c, ["--pmc-child", ",".join(a.items), "--points", str(a.points), "--history", "packed", "--workload", "von_mises_mixed"], 400)
    if res is None:
        print(f"# {c}: pass failed", flush=True)
        continue
    child, vals = res
    table[c] = slice_timed(child.get("log", []), vals)
    print(f"# {c}: {len(vals)} dispatches, {len(table[c])} items", flush=True)
os.makedirs(os.path.dirname(a.out), exist_ok=True)
with open(a.out, "w") as f:
    f.write(f"| counter (per launch, avg of 4 timed launches, n = {a.points}) | " + " | ".join(a.items) + " |\n|---|" + "---|" * len(a.items) + "\n")
    for c, row in table.items():
        f.write(f"| `{c}` | " + " | ".join(f"{row[i]:.4g}" if i in row else "--" for i in a.items) + " |\n")
    rd, wr, w64 = table.get("TCC_EA0_RDREQ_sum", {}), table.get("TCC_EA0_WRREQ_sum", {}), table.get("TCC_EA0_WRREQ_64B_sum", {})
    r32, r64, r128 = table.get("TCC_EA0_RDREQ_32B_sum", {}), table.get("TCC_EA0_RDREQ_64B_sum", {}), table.get("TCC_EA0_RDREQ_128B_sum", {})
    f.write("| write bytes by request size: 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B), GB | "
            + " | ".join(f"{(64 * w64[i] + 32 * (wr[i] - w64[i])) / 1e9:.2f}" if (i in wr and i in w64) else "--" for i in a.items) + " |\n")
    f.write("| share of the write requests that are 64-byte ones | " + " | ".join(f"{w64[i] / wr[i]:.3f}" if (i in wr and i in w64 and wr[i]) else "--" for i in a.items) + " |\n")
    f.write("| read bytes by request size: 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B, GB | "
            + " | ".join(f"{(32 * r32[i] + 64 * r64[i] + 128 * r128[i]) / 1e9:.2f}" if (i in r32 and i in r64 and i in r128) else "--" for i in a.items) + " |\n")
print(open(a.out).read())
