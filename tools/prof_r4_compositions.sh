#!/bin/bash
# rocprofv3 kernel-trace summaries of the other BASELINE shapes on one GPU (round 4): the mixed batch (configs[4]'s share), the strips
# form of the banded solver (configs[2]), the block-angular compute (configs[3]).  Usage: bash tools/prof_r4_compositions.sh OUTDIR
set -u
OUT=${1:-gpurun_out/prof_comp4}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/mixed" -- python3 "$ROOT/tools/mixed_only.py" 12500 > "$ROOT/$OUT/mixed.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/strips" -- python3 "$ROOT/tools/strips_probe.py" 1024 > "$ROOT/$OUT/strips.log" 2>&1
QRK_BIG=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/angular" -- python3 "$ROOT/tools/angular_probe.py" > "$ROOT/$OUT/angular.log" 2>&1
cd "$ROOT"
for d in mixed strips angular; do
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { echo "## $d"; cut -c1-200 "$f" | head -14; } >> "$OUT/kernel_stats_all.txt"
  find "$OUT/$d" -name "*.db" -delete 2>/dev/null; find "$OUT/$d" -name "*trace.csv" -delete 2>/dev/null
done
cat "$OUT/kernel_stats_all.txt"
