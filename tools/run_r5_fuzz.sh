#!/bin/bash
# round 5: fuzz of the kernels that changed (bdqr_pair4 in both forms, bdqr_w64, bdqr_reg) against the oracle
OUT=gpurun_out/r5fuzz
mkdir -p $OUT
timeout -k 10 500 python tools/fuzz_pair4.py 200 5000 > $OUT/fuzz_pair4.txt 2>&1; tail -3 $OUT/fuzz_pair4.txt
QRK_P4_OWN=1 timeout -k 10 500 python tools/fuzz_pair4.py 200 6000 > $OUT/fuzz_pair4_own.txt 2>&1; tail -3 $OUT/fuzz_pair4_own.txt
timeout -k 10 500 python tools/fuzz_w64.py 150 5000 > $OUT/fuzz_w64.txt 2>&1; tail -3 $OUT/fuzz_w64.txt
timeout -k 10 600 python tools/fuzz_onchip.py 60 5000 > $OUT/fuzz_onchip.txt 2>&1; tail -3 $OUT/fuzz_onchip.txt
