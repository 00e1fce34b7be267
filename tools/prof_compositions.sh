#!/bin/bash
# rocprofv3 kernel-trace summaries of the banded chain and of a mixed-size batch (GPU box).  Usage: bash tools/prof_compositions.sh OUTDIR
set -u
OUT=${1:-gpurun_out/prof_comp}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/banded" -- python3 "$ROOT/tools/banded_probe.py" 512 > "$ROOT/$OUT/banded.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/mixed" -- python3 "$ROOT/tools/mixed_probe.py" 4000 > "$ROOT/$OUT/mixed.log" 2>&1
cd "$ROOT"
for d in banded mixed; do
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cut -c1-220 "$f" | head -25 > "$OUT/${d}_kernel_stats.csv"
done
ls -la "$OUT"
