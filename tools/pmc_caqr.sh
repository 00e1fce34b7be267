#!/bin/bash
# MFMA counters of the two-stage dense factorisation (caqr.hip), separate rocprofv3 --pmc passes (counters only).  GPU box.
set -u
OUT=${1:-gpurun_out/pmc_caqr}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|MOPS" | head -60 > "$ROOT/$OUT/mfma_counters_available.txt"
i=0
for grp in "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/p$i" -- python3 "$ROOT/tools/caqr_probe.py" > "$ROOT/$OUT/p$i.log" 2>&1 || echo "pass $i failed: $grp"
done
ls -R "$ROOT/$OUT" | head -40
