#!/bin/bash
# End-of-round evidence on the GPU box: full GPU suite, bench line, rocprofv3 kernel stats of the same bench command, composition timings.
OUT=gpurun_out/r3end
mkdir -p $OUT
ROOT=$(pwd)
timeout -k 10 1100 python -m pytest tests -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "rc=$?" >> $OUT/gpu_tests.txt
tail -3 $OUT/gpu_tests.txt
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-steady --no-e2e --no-other > $ROOT/$OUT/prof.log 2>&1
cd $ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -8 > $OUT/kernel_stats.csv; cat $OUT/kernel_stats.csv
tail -1 $OUT/prof.log | cut -c1-400 > $OUT/bench_profiled_line.txt
find $OUT/prof -name "*.db" -delete 2>/dev/null; find $OUT/prof -name "*trace.csv" -delete 2>/dev/null
timeout -k 10 300 python tools/mixed_probe.py 4000 2>&1 | grep tiles/s > $OUT/mixed_probe.txt; head -3 $OUT/mixed_probe.txt
timeout -k 10 200 python tools/k2_wgs_probe.py 0 2>&1 | grep tiles/s > $OUT/k2_sizes.txt; cat $OUT/k2_sizes.txt
QRK_BIG=1 timeout -k 10 300 python tools/angular_probe.py 2>&1 | grep compute > $OUT/angular_probe.txt; cat $OUT/angular_probe.txt
timeout -k 10 200 python tools/caqr_probe.py 2>&1 | grep factorize > $OUT/caqr_probe.txt; cat $OUT/caqr_probe.txt
timeout -k 10 300 python tools/strips_probe.py 2048 2>&1 | grep strips > $OUT/strips_probe.txt; cat $OUT/strips_probe.txt
