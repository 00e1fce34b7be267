#!/bin/bash
# MFMA counters of K2 (mid-size tiles, Q accumulation on MFMA) and K4 (banded chain: block reflector on MFMA): separate --pmc passes.
set -u
OUT=${1:-gpurun_out/pmc_k2k4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for probe in "mixed_probe.py 2000" "banded_probe.py 256"; do
  name=$(echo $probe | cut -d. -f1); script=$(echo $probe | cut -d" " -f1); arg=$(echo $probe | cut -d" " -f2)
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d "$ROOT/$OUT/${name}_sq" -- python3 "$ROOT/tools/$script" $arg > "$ROOT/$OUT/${name}_sq.log" 2>&1 || echo "failed $probe sq"
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$ROOT/$OUT/${name}_grbm" -- python3 "$ROOT/tools/$script" $arg > "$ROOT/$OUT/${name}_grbm.log" 2>&1 || echo "failed $probe grbm"
done
ls -R "$ROOT/$OUT" | grep counter_collection
