#!/bin/bash
# Round-3 counter evidence (GPU box): K1 dynamic instruction mix + HBM traffic on the bench command; K2 (256 x 256 tiles) MFMA use and
# fetch volume.  Counters only, one group per rocprofv3 pass (no trace domains next to --pmc).
set -u
OUT=${1:-gpurun_out/r3pmc}
WHAT=${2:-all}            # k1 | k2 | all
K2KERNEL=${3:-bdqr_reg}   # kernel whose dispatches are averaged in the K2 part (bdqr_col with QRK_COL_ONCHIP=0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT/k1" "$ROOT/$OUT/k2"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 16 --warmup 4 --no-cpu-baseline --no-steady --no-check --no-e2e --no-other"
i=0
if [ "$WHAT" != k2 ]; then
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/k1/p$i" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/k1/p$i.log" 2>&1 || echo "K1 pass $i failed: $grp"
  echo "K1 pass $i done"
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT/k1" bdqr_pair32 > "$ROOT/$OUT/k1_summary.txt" 2>&1
fi
[ "$WHAT" = k1 ] && { cat "$ROOT/$OUT/k1_summary.txt"; exit 0; }
i=0
for grp in "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/k2/p$i" -- python3 "$ROOT/tools/k2_256_probe.py" 1024 2 > "$ROOT/$OUT/k2/p$i.log" 2>&1 || echo "K2 pass $i failed: $grp"
  echo "K2 pass $i done"
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT/k2" $K2KERNEL > "$ROOT/$OUT/k2_summary.txt" 2>&1
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/k2/trace" -o k2 -- python3 "$ROOT/tools/k2_256_probe.py" 1024 2 > "$ROOT/$OUT/k2/trace.log" 2>&1
f=$(find "$ROOT/$OUT/k2/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -8 > "$ROOT/$OUT/k2_kernel_stats.csv"
cat "$ROOT/$OUT/k2_summary.txt" "$ROOT/$OUT/k2_kernel_stats.csv"
# keep what travels back small
find "$ROOT/$OUT" -name "*.db" -delete 2>/dev/null
find "$ROOT/$OUT" -name "*agent_info.csv" -delete 2>/dev/null
