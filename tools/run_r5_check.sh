#!/bin/bash
# round 5 checkpoint: full GPU suite + the bench line as the driver runs it
OUT=${1:-gpurun_out/r5chk}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $OUT/gpu_tests.txt 2>&1; echo "rc=$?" >> $OUT/gpu_tests.txt
tail -4 $OUT/gpu_tests.txt
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
