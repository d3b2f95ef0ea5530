#!/bin/bash
# VALU / LDS / SALU instructions per launch of the evaluate kernels of one bench workload under several builds (one --pmc pass each):
#     tools/valu_count.sh <workload> <lib.so | ship> ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
    tag=$(basename "$L" .so)
    if [ "$L" = ship ]; then unset FCAMD_LIBRARY; else export FCAMD_LIBRARY=$R/$L; fi
    OUT=$R/gpurun_out/prof/valu_$tag; rm -rf "$OUT"; mkdir -p "$OUT"
    timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d "$OUT/q" -- python3 "$R/bench.py" --workload "$W" --steps 4 --warmup 1 --no-cpu-baseline --configs none --no-host-path --no-live-traffic --placement-tries 1 > "$OUT/log" 2>&1
    python3 - "$OUT" "$tag" <<'PY'
import csv, glob, sys
from collections import defaultdict
v = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/q/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fcamd::evaluate" in r["Kernel_Name"]:
            v[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in v.items():
    print(sys.argv[2], k, {n: round(sum(x) / len(x) / 1e6, 1) for n, x in c.items()}, "M per launch,", len(next(iter(c.values()))), "launches")
PY
done
