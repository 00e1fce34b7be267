#!/bin/bash
# Collect PMC counters for the factorisation kernel in separate rocprofv3 passes (counters only:
# no --sys-trace/--hip-trace alongside --pmc).  Usage (on the GPU box, via gpurun):
#   bash tools/pmc_passes.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 16 --warmup 4 --no-cpu-baseline --no-steady --no-check $*"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_IFETCH SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/p$i" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/p$i.log" 2>&1 || echo "pass $i failed: $grp"
done
# QRK_PMC_KERNEL: substring of the kernel name to summarise (default: every bdqr kernel of the run); the raw pass directories are
# removed afterwards unless QRK_PMC_KEEP is set (gpurun merges at most 64 MiB back)
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" ${QRK_PMC_KERNEL:-bdqr} > "$ROOT/$OUT/summary.txt" 2>&1
[ -n "${QRK_PMC_KEEP:-}" ] || rm -rf "$ROOT/$OUT"/p[0-9]*
cat "$ROOT/$OUT/summary.txt"
