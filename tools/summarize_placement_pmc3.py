#!/usr/bin/env python3
"""Per dispatch of the evaluate kernel: duration and the counters of each tools/placement_pmc3.sh pass."""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
for d in sorted(glob.glob(os.path.join(src, "p_*"))):
    if not os.path.isdir(d):
        continue
    f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")))
    if not f:
        continue
    per = defaultdict(dict)
    for r in csv.DictReader(open(f[-1])):
        if "fcamd::evaluate_kernel" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        per[k]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        per[k][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[1:]  # drop the in-place warm step
    names = [n for n in per[ids[0]] if n != "ms"]
    print(os.path.basename(d), names)
    # 5 dispatches per candidate (1 warm + 4 timed): average the last four
    for c in range(len(ids) // 5):
        grp = [per[i] for i in ids[5 * c + 1: 5 * c + 5]]
        ms = sum(g["ms"] for g in grp) / len(grp)
        vals = [sum(g[n] for g in grp) / len(grp) for n in names]
        print(f"  cand {c}: {ms:7.3f} ms  " + "  ".join(f"{v:14.4g}" for v in vals))
