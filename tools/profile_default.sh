#!/bin/bash
# Run on the GPU box: `rocprofv3 --kernel-trace --stats` of the DEFAULT bench command (python3 bench.py; --detail only names where the
# full record goes -- the launch logs the summary slices the trace with).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof/default_bench
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" --detail "$OUT/bench_under_rocprof.json" > "$OUT/kt.log" 2>&1
echo "kt exit $?" >> "$OUT/kt.log"
grep -h '"metric"' "$OUT/kt.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
