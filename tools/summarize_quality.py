#!/usr/bin/env python3
"""Summarise tools/profile_quality.sh output: per counter, the average over the timed launches of the
evaluate kernel.  Usage: summarize_quality.py gpurun_out/prof/<tag> <round> <workload>  -> appends a
section to profiles/r<round>_<workload>_rocprof.md"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, rnd, workload = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, "q_*", "*", "*_counter_collection.csv")), key=os.path.getmtime):
    rows = [r for r in csv.DictReader(open(f)) if "fcamd::evaluate_kernel" in r["Kernel_Name"]]
    by_counter = defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in by_counter.items():
        vals[k] = v[1:] if len(v) > 1 else v  # drop the in-place warm step
out = os.path.join(ROOT, "profiles", f"r{rnd}_{workload}_rocprof.md")
with open(out, "a") as f:
    f.write("\n## kernel-quality counters (tools/profile_quality.sh, one `--pmc` pass per group; average per launch of the evaluate kernel)\n\n| counter | value |\n|---|---|\n")
    for k in sorted(vals):
        v = vals[k]
        f.write(f"| {k} | {sum(v) / len(v):.4g} |\n")
print(open(out).read()[-1500:])
