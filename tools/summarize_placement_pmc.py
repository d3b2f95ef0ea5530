#!/usr/bin/env python3
"""Join the per-dispatch counters of tools/vmm_placement_pmc.sh with the placements of the probe:

    python tools/summarize_placement_pmc.py gpurun_out/prof/vmm_placement [out.md]

The probe prints one JSON line per placement, in dispatch order; every placement is 5 dispatches of the
evaluate kernel (1 warm + 4 timed) after the single in-place warm step of the set-up."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
table = defaultdict(dict)  # placement -> column -> value
order = []
for d in sorted(glob.glob(os.path.join(src, "p_*"))):
    if not os.path.isdir(d):
        continue
    f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")))
    log = d + ".log"
    if not f or not os.path.exists(log):
        continue
    labels = []
    for line in open(log):
        line = line.strip()
        if line.startswith("{") and '"placement"' in line:
            labels.append(json.loads(line))
    per = defaultdict(dict)
    for r in csv.DictReader(open(f[-1])):
        if "fcamd::evaluate_kernel" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        per[k]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        per[k][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[1:]  # drop the in-place warm step
    names = [n for n in per[ids[0]] if n != "ms"]
    for c, lab in enumerate(labels):
        grp = [per[i] for i in ids[5 * c + 1: 5 * c + 5]]
        if len(grp) < 4:
            break
        p = lab["placement"]
        if p not in order:
            order.append(p)
                table[p]["ms under --pmc " + os.path.basename(d)] = round(sum(g["ms"] for g in grp) / len(grp), 3)
        for nme in names:
            table[p][nme] = sum(g[nme] for g in grp) / len(grp)
cols = []
for p in order:
    for c in table[p]:
        if c not in cols:
            cols.append(c)
out.write("| placement | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols) + "\n")
for p in order:
    out.write(f"| {p} | " + " | ".join((f"{table[p][c]:.4g}" if isinstance(table[p].get(c), float) else str(table[p].get(c, ""))) for c in cols) + " |\n")
