#!/bin/bash
# Same counter set over many processes: correlate counters with the per-process kernel time.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/placement2
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SET=${SET:-"TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_BUSY"}
for i in 1 2 3 4 5 6 7 8; do
  mode=separate; [ $((i % 2)) -eq 0 ] && mode=slab
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d "$OUT/run_$i" -- python3 "$R/tools/placement_pmc.py" $mode > "$OUT/run_$i.log" 2>&1
done
python3 - <<PY
import csv, glob
for d in sorted(glob.glob("$OUT/run_*/")):
    f = glob.glob(d + "*/*_counter_collection.csv")
    if not f: continue
    rows = [r for r in csv.DictReader(open(f[0])) if "fcamd::evaluate_kernel" in r["Kernel_Name"]]
    last = {}
    for r in rows:
        last.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        last[r["Dispatch_Id"]]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    k = sorted(last, key=int)[-1]
    mode = open(d.rstrip("/") + ".log").read().split()[0] if True else ""
    print(d.split("/")[-2], {kk: (round(v, 3) if kk == "ms" else int(v)) for kk, v in last[k].items()})
PY
