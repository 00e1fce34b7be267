#!/bin/bash
# round 5: kernel trace (timestamps) of one 40 000 x 2 000 two-stage factorisation
OUT=$PWD/gpurun_out/r5k3
mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k3 -- python3 $ROOT/tools/caqr_probe.py > $OUT/log.txt 2>&1
cd $ROOT
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python3 tools/k3_timeline.py $f > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
head -1 $f

cp $f $OUT/k3_kernel_trace.csv; find $OUT/prof -name "*.db" -delete 2>/dev/null; find $OUT/prof -name "*trace.csv" -delete 2>/dev/null
