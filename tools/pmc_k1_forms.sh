#!/bin/bash
# round 6: instruction counters of ONE 10 000-tile launch of the two two-phase 32 x 32 kernels (the headline run of bench.py without its other
# legs: every dispatch of the kernel is a 10 000-tile launch).  Usage (GPU box): bash tools/pmc_k1_forms.sh gpurun_out/pmc_forms
OUT=${1:-gpurun_out/pmc_forms}
ROOT=$(pwd)
mkdir -p $OUT
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-steady --no-e2e --no-other --no-check"
cd /tmp && export TMPDIR=/tmp
for form in pair4 quad32; do
  export QRK_K1_FORM=$form
  mkdir -p "$ROOT/$OUT/$form"
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/$form/p$i" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/$form/p$i.log" 2>&1 || echo "PMC pass $i failed: $grp"
  done
  python3 $ROOT/tools/pmc_summary.py $ROOT/$OUT/$form bdqr_$form > $ROOT/$OUT/${form}_summary.txt 2>&1
  rm -rf $ROOT/$OUT/$form
  echo "== $form"; cat $ROOT/$OUT/${form}_summary.txt
done
