#!/bin/bash
# rocprofv3 kernel-trace summaries of the small-tile kernel probe and of bench.py (GPU box).  Usage: bash tools/prof_small.sh OUTDIR
set -u
OUT=${1:-gpurun_out/prof_small}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/small" -- python3 "$ROOT/tools/small_probe_big.py" > "$ROOT/$OUT/small.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/bench" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/$OUT/bench.log" 2>&1
cd "$ROOT"
for d in small bench; do
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cut -c1-220 "$f" | head -25 > "$OUT/${d}_kernel_stats.csv"
done
ls -la "$OUT"
