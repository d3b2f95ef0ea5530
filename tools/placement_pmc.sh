#!/bin/bash
# Run on the GPU box: PMC passes over the slab / separate placements (see tools/placement_pmc.py).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/placement
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in slab separate; do
  i=0
  for set in "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY" \
             "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_BUSY" \
             "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --output-format csv -d "$OUT/${mode}_$i" -- python3 "$R/tools/placement_pmc.py" $mode > "$OUT/${mode}_$i.log" 2>&1
    grep "kernel ms" "$OUT/${mode}_$i.log"
  done
done
