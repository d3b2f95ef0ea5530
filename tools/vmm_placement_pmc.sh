#!/bin/bash
# Run on the GPU box: memory-side PMC counters per dispatch while tools/vmm_placement_probe.py walks through
# its placements (torch.empty candidates vs VMM working sets).  One --pmc pass per group (no tracing options
# next to --pmc); summarise with tools/summarize_placement_pmc.py gpurun_out/prof/vmm_placement
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof/vmm_placement
N=${1:-5e7}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_BUSY_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p_$i" -- python3 "$R/tools/vmm_placement_probe.py" "$N" 6 > "$OUT/p_$i.log" 2>&1
  echo "$grp exit $?" >> "$OUT/p_$i.log"
done
