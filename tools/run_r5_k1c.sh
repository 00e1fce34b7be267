#!/bin/bash
# round 5, K1 experiments, third batch: the combinations, the search of step K + 1 between the update FMAs of step K; parity of the candidate
OUT=gpurun_out/r5k1c
mkdir -p $OUT
QRK_AB_HASH=1 timeout -k 10 300 python tools/ab.py run 10000 > $OUT/ab_10000.txt 2>&1; tail -7 $OUT/ab_10000.txt
timeout -k 10 300 python tools/ab.py run 100000 > $OUT/ab_100000.txt 2>&1; tail -6 $OUT/ab_100000.txt
timeout -k 10 300 python tools/ab.py run 1250 > $OUT/ab_1250.txt 2>&1; tail -6 $OUT/ab_1250.txt
timeout -k 10 300 python tools/ab.py run 20000 > $OUT/ab_20000.txt 2>&1; tail -6 $OUT/ab_20000.txt
QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_combo4a.so timeout -k 10 600 python -m pytest tests/test_pair_generations_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py tests/test_bd_gpu.py tests/test_golden_gpu.py -q -m gpu -x > $OUT/tests_combo4a.txt 2>&1; tail -5 $OUT/tests_combo4a.txt
timeout -k 10 300 python -m pytest tests/test_banded_strips_gpu.py -q -m gpu -x -k "not full" > $OUT/tests_strips.txt 2>&1; tail -5 $OUT/tests_strips.txt
