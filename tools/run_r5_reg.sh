#!/bin/bash
# round 5: bdqr_reg A/B (tools/abl: old = HEAD, new = integer arg-max fast paths), per-phase ticks, parity
OUT=gpurun_out/r5reg
mkdir -p $OUT
for v in old new old new; do echo "== $v"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/k2_wgs_probe.py one; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_newprof.so timeout -k 10 100 python tools/k2_256_probe.py 256 1 2>&1 | grep -E "prof|search" | tail -4 > $OUT/prof.txt; cat $OUT/prof.txt
QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_new.so timeout -k 10 600 python -m pytest tests/test_onchip_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py -q -m gpu -x 2>&1 | tail -3 > $OUT/tests.txt; cat $OUT/tests.txt
QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_new.so timeout -k 10 300 python tools/mixed_probe.py 4000 2>&1 | grep "mixed" > $OUT/mixed.txt; cat $OUT/mixed.txt
