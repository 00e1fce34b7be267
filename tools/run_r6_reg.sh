#!/bin/bash
# round 6: bdqr_reg A/B (tools/abl: old = the committed kernel, new = the working tree), per-phase ticks, parity, the mixed batch
OUT=gpurun_out/r6reg
mkdir -p $OUT
for v in old new old new; do echo "== $v"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/k2_wgs_probe.py one; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_newprof.so timeout -k 10 100 python tools/k2_256_probe.py 256 1 2>&1 | grep -E "prof|search" | tail -4 > $OUT/prof.txt; cat $OUT/prof.txt
QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_new.so timeout -k 10 600 python -m pytest tests/test_onchip_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py -q -m gpu -x 2>&1 | tail -3 > $OUT/tests.txt; cat $OUT/tests.txt
for v in old new; do echo "== $v"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 300 python tools/mixed_only.py 12500 2>&1 | grep "mixed"; done > $OUT/mixed.txt; cat $OUT/mixed.txt
