#!/bin/bash
# round 5: the 8-row instantiation of bdqr_quad (tiles of 5..8 rows, eight per wave, two tiles per DPP row) against bdqr_small's groups of 8 lanes
OUT=gpurun_out/r5quad8
mkdir -p $OUT
for m in 9 5 9 5; do echo "== QRK_QUAD_MIN_ROWS=$m (9: bdqr_small for these shapes, 5: bdqr_quad<8>)"; QRK_QUAD_MIN_ROWS=$m timeout -k 10 200 python tools/quad_probe.py small 2>&1 | grep " x "; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout -k 10 600 python -m pytest tests/test_quad_gpu.py tests/test_small_tiles_gpu.py tests/test_golden_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py tests/test_bd_gpu.py tests/test_lm_gpu.py tests/test_angular.py -q -m gpu -x 2>&1 | tail -4 > $OUT/tests.txt; cat $OUT/tests.txt
