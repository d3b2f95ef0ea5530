#!/bin/bash
# headline step of bench.py under several ENVIRONMENTS (launch knobs), one process each, the six hipMalloc candidates of the tangent listed:
#     tools/ab_env.sh <out-dir> "NAME=VALUE ..." "NAME=VALUE ..." ...     ("-" = no extra environment)
OUT=$1; shift
mkdir -p "$OUT"
i=0
for E in "$@"; do
    i=$((i+1))
    tag="run$i"
    if [ "$E" = "-" ]; then envs=""; else envs="$E"; fi
    env $envs python bench.py --no-host-path --configs none --no-live-traffic --no-cpu-baseline --steps 6 --placement tune ${AB_ARGS:-} > "$OUT/$tag.json" 2> "$OUT/$tag.err" || tail -3 "$OUT/$tag.err"
    python - "$OUT/$tag.json" "$E" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    c = sorted(d["placement"]["hipmalloc_tangent_candidate_ms"])
    print(f"{sys.argv[2]:>24}: timed {r['kernel_ms_avg']:.3f} ms frac {r['frac']:.4f}; candidates min {c[0]:.3f} med {c[len(c)//2]:.3f} max {c[-1]:.3f}  first {d['placement']['hipmalloc_tangent_candidate_ms'][0]:.3f}")
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done
