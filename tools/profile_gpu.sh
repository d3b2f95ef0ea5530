#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + the two PMC passes for one
# bench workload, as MI355X_MICROARCH.md prescribes (separate --pmc passes; FETCH_SIZE and
# WRITE_SIZE do not fit one pass).  Usage: tools/profile_gpu.sh <tag> --workload <name> [bench args...]
# Writes gpurun_out/prof/<tag>/{kt,fetch,write}/... ; summarise with tools/summarize_profile.py.
set -u
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --detail "$OUT/bench_under_rocprof.json" "$@" > "$OUT/kt.log" 2>&1
echo "kt exit $?" >> "$OUT/kt.log"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --detail "$OUT/fetch_detail.json" "$@" > "$OUT/fetch.log" 2>&1
echo "fetch exit $?" >> "$OUT/fetch.log"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --detail "$OUT/write_detail.json" "$@" > "$OUT/write.log" 2>&1
echo "write exit $?" >> "$OUT/write.log"
# (the full record of each run -- launch logs included -- is its --detail file; the stdout line is the compact one)
