#!/bin/bash
# rocprofv3 kernel stats of the two-kernel 32x32 path.  Usage: bash tools/prof_split.sh OUTDIR [LIB]
OUT=${1:-gpurun_out/prof_split}; LIB=${2:-}
ROOT=$(pwd); mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
export QRK_SPLIT=1
[ -n "$LIB" ] && export QRKIT_AMD_LIB="$ROOT/$LIB"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 100 --warmup 10 > "$ROOT/$OUT.log" 2>&1
cd "$ROOT"; f=$(find "$OUT" -name "*kernel_stats.csv" | head -1); cut -c1-120 "$f" | head -3
