#!/bin/bash
# HBM traffic and issue counters of the bench command by PMC (counter-only passes, one group each), round 4.
OUT=${1:-gpurun_out/r4pmc}
ROOT=$(pwd)
mkdir -p $OUT/pmc
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-steady --no-e2e --no-other --no-check"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/pmc/p$i" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/pmc/p$i.log" 2>&1 || echo "PMC pass $i failed: $grp"
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc bdqr_pair4 > $OUT/k1_pmc_summary.txt 2>&1; cat $OUT/k1_pmc_summary.txt
find $OUT/pmc -name "*.db" -delete 2>/dev/null; find $OUT/pmc -name "*agent_info.csv" -delete 2>/dev/null
