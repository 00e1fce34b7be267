#!/bin/bash
# round 5: bdqr_w64 with the four-waves-per-SIMD instantiation for tiles of 33..44 rows; QRK_W64_WPS=3 is the round-4 occupancy
OUT=gpurun_out/r5w64b
mkdir -p $OUT
for w in 3 0 3 0; do echo "== QRK_W64_WPS=$w (0: default, four waves per SIMD up to 44 rows)"; QRK_W64_WPS=$w timeout -k 10 200 python tools/w64_small_batches.py 2>&1 | grep -E "^ (33|40|44|48) x" ; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout -k 10 600 python -m pytest tests/test_w64_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py tests/test_onchip_gpu.py tests/test_bd_gpu.py -q -m gpu -x 2>&1 | tail -2 > $OUT/tests.txt; cat $OUT/tests.txt
timeout -k 10 300 python tools/fuzz_w64.py 150 7000 2>&1 | tail -1 >> $OUT/tests.txt; tail -1 $OUT/tests.txt
