#!/bin/bash
# round 5: which kernels the solve() of BASELINE configs[3] spends its 8 ms in
OUT=$PWD/gpurun_out/r5angsolve
mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o a -- python3 $ROOT/tools/angular_solve_prof.py > $OUT/log.txt 2>&1
cd $ROOT
tail -2 $OUT/log.txt
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cut -c1-200 $f | head -40 > $OUT/kernel_stats.csv; cat $OUT/kernel_stats.csv
find $OUT/prof -name "*.db" -delete 2>/dev/null; find $OUT/prof -name "*trace.csv" -delete 2>/dev/null
