#!/bin/bash
# End-of-round evidence on the GPU box (round 6): full GPU suite, the bench line as the driver runs it, rocprofv3 kernel stats of the same
# bench command, HBM traffic by PMC (separate counter-only passes), size sweep and composition timings.
OUT=${1:-gpurun_out/r6end}
mkdir -p $OUT
ROOT=$(pwd)
timeout -k 10 1100 python -m pytest tests -q -m gpu > $OUT/gpu_tests.txt 2>&1; echo "rc=$?" >> $OUT/gpu_tests.txt
tail -3 $OUT/gpu_tests.txt
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-steady --no-e2e --no-other --no-check"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python3 $ROOT/bench.py $ARGS > $ROOT/$OUT/prof.log 2>&1
cd $ROOT
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -8 > $OUT/kernel_stats.csv; cat $OUT/kernel_stats.csv
tail -1 $OUT/prof.log | cut -c1-600 > $OUT/bench_profiled_line.txt
find $OUT/prof -name "*.db" -delete 2>/dev/null; find $OUT/prof -name "*trace.csv" -delete 2>/dev/null
mkdir -p $OUT/pmc
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$OUT/pmc/p$i" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/pmc/p$i.log" 2>&1 || echo "PMC pass $i failed: $grp"
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc bdqr_pair4 > $OUT/k1_pmc_summary.txt 2>&1; cat $OUT/k1_pmc_summary.txt
find $OUT/pmc -name "*.db" -delete 2>/dev/null; find $OUT/pmc -name "*agent_info.csv" -delete 2>/dev/null
timeout -k 10 300 python tools/mixed_probe.py 4000 2>&1 | grep tiles/s > $OUT/mixed_probe.txt; head -14 $OUT/mixed_probe.txt
timeout -k 10 200 python tools/k2_wgs_probe.py 0 2>&1 | grep tiles/s > $OUT/k2_sizes.txt; cat $OUT/k2_sizes.txt
QRK_BIG=1 timeout -k 10 300 python tools/angular_probe.py 2>&1 | grep compute > $OUT/angular_probe.txt; cat $OUT/angular_probe.txt
timeout -k 10 200 python tools/caqr_probe.py 2>&1 | grep factorize > $OUT/caqr_probe.txt; cat $OUT/caqr_probe.txt
timeout -k 10 300 python tools/strips_probe.py 2048 2>&1 | grep strips > $OUT/strips_probe.txt; cat $OUT/strips_probe.txt
