#!/usr/bin/env python3
"""Join the passes of tools/placement_counters.sh: per counter, its value on the best and on the worst draw of the pass.
    python tools/summarize_placement_counters.py <dir with pass_<COUNTER>/...> > profiles/rNN_placement_counters.md"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
print("| counter | pass: candidate ms (best / worst) | best draw (avg of 4 launches) | worst draw | worst / best |\n|---|---|---|---|---|")
for p in sorted(glob.glob(os.path.join(d, "pass_*"))):
    counter = os.path.basename(p)[5:]
    try:
        meta = json.load(open(os.path.join(p, "marks.json")))
        f = sorted(glob.glob(os.path.join(p, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
        rows = [r for r in csv.DictReader(open(f)) if "fcamd::evaluate" in r["Kernel_Name"] and r.get("Counter_Name", counter) == counter]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals = [float(r["Counter_Value"]) for r in rows]
        avg = {k: sum(vals[i] for i in idx) / len(idx) for k, idx in meta["evaluate_dispatch_index"].items()}
        ms = meta["candidate_ms"]
        print(f"| `{counter}` | {ms[meta['best']]:.3f} / {ms[meta['worst']]:.3f} | {avg['best']:.4g} | {avg['worst']:.4g} | {avg['worst'] / max(avg['best'], 1e-30):.2f} |")
    except Exception as e:  # noqa: BLE001
        print(f"| `{counter}` | failed: {e} | | | |")
