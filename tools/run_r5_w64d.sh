#!/bin/bash
# round 5: bdqr_w64 without tau in LDS (17 920 B per wave at 64 rows: NINE waves per CU instead of eight) against the build before
OUT=gpurun_out/r5w64d
mkdir -p $OUT
for v in old new old new; do echo "== $v"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/w64_small_batches.py 2>&1 | grep -E "^ (48|56|64) x" | grep -E "B=  (2000|4096)|B= 20000"; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout -k 10 600 python -m pytest tests/test_w64_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py tests/test_bd_gpu.py -q -m gpu -x 2>&1 | tail -2 > $OUT/tests.txt; cat $OUT/tests.txt
timeout -k 10 300 python tools/fuzz_w64.py 150 31000 2>&1 | tail -1
