#!/bin/bash
# Run on the GPU box (through gpurun): kernel-quality counters for one bench workload, one --pmc
# pass per counter group (no tracing options next to --pmc).  Usage: tools/profile_quality.sh <tag> [bench args]
# Writes gpurun_out/prof/<tag>/q_<group>/...; summarise with tools/summarize_quality.py.
set -u
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "LDSBankConflict" "OccupancyPercent" "VALUBusy" "MemUnitBusy" "MemUnitStalled" "WriteUnitStalled" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/q_$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$OUT/q_$i.log" 2>&1
  echo "$grp exit $?" >> "$OUT/q_$i.log"
done
