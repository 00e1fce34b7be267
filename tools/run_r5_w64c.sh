#!/bin/bash
# round 5: bdqr_w64's backward steps with the reflector of step k - 1 fetched before the FMAs of step k (new) against the build before (old)
OUT=gpurun_out/r5w64c
mkdir -p $OUT
for v in old new old new; do echo "== $v"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/w64_small_batches.py 2>&1 | grep -E "B=  (2000|4096)|B= 20000"; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout -k 10 600 python -m pytest tests/test_w64_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py -q -m gpu -x 2>&1 | tail -2 > $OUT/tests.txt; cat $OUT/tests.txt
timeout -k 10 300 python tools/fuzz_w64.py 150 7000 2>&1 | tail -2
