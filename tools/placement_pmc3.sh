#!/bin/bash
# Run on the GPU box: memory-side PMC counters per dispatch of the evaluate kernel while the tangent lives in
# eight candidate allocations (tools/tangent_placement_probe.py) -- which counter separates slow from fast
# placements?  One --pmc pass per group (no tracing options next to --pmc); summarise with
# tools/summarize_placement_pmc3.py gpurun_out/prof/placement3
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof/placement3
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum" \
           "TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p_$i" -- python3 "$R/tools/tangent_placement_probe.py" 5e7 8 > "$OUT/p_$i.log" 2>&1
  echo "$grp exit $?" >> "$OUT/p_$i.log"
done
