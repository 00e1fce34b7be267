#!/bin/bash
# rocprofv3 kernel-trace summary of the block-angular composition at the BASELINE configs[3] shape (GPU box).
set -u
OUT=${1:-gpurun_out/prof_ang}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export QRK_BIG=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/ang" -- python3 "$ROOT/tools/angular_probe.py" > "$ROOT/$OUT/ang.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/ang" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-200 "$f" | head -30 > "$OUT/angular_kernel_stats.csv"
tail -5 "$OUT/ang.log"
