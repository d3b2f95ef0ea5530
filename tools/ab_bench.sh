#!/bin/bash
# headline step of bench.py under several builds of the library, one process each (placement tuned in every process):
#     tools/ab_bench.sh <out-dir> <lib.so | ship> ...      e.g. gpurun -- 'tools/ab_bench.sh gpurun_out/ab ship tools/_ab/libfcamd_x.so'
OUT=$1; shift
mkdir -p "$OUT"
for L in "$@"; do
    tag=$(basename "$L" .so)
    if [ "$L" = ship ]; then unset FCAMD_LIBRARY; else export FCAMD_LIBRARY=$PWD/$L; fi
    python bench.py --no-host-path --configs none --no-live-traffic --no-cpu-baseline --steps 10 ${AB_ARGS:-} > "$OUT/$tag.json" 2> "$OUT/$tag.err" || { tail -3 "$OUT/$tag.err"; }
    python - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:>16}: kernel {r['kernel_ms_avg']:.3f} ms (min {r['kernel_ms_min']:.3f}) frac {r['frac']:.4f}  first {r.get('frac_first_allocation')} worst {r.get('frac_worst_candidate')}  "
          f"unpacked {d.get('sparse_unpacked_history', {}).get('kernel_ms_avg')} full {d.get('full_trial_history', {}).get('kernel_ms_avg')}")
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done
