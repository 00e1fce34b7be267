#!/bin/bash
# round 5, final build: longer fuzz of every kernel family against the oracle (fresh seeds)
OUT=gpurun_out/r5fuzz2
mkdir -p $OUT
timeout -k 10 400 python tools/fuzz_pair4.py 400 20000 > $OUT/fuzz_pair4.txt 2>&1; tail -1 $OUT/fuzz_pair4.txt
QRK_P4_OWN=1 timeout -k 10 400 python tools/fuzz_pair4.py 300 21000 > $OUT/fuzz_pair4_own.txt 2>&1; tail -1 $OUT/fuzz_pair4_own.txt
timeout -k 10 400 python tools/fuzz_quad.py 1500 22000 > $OUT/fuzz_quad.txt 2>&1; tail -1 $OUT/fuzz_quad.txt
timeout -k 10 400 python tools/fuzz_w64.py 300 23000 > $OUT/fuzz_w64.txt 2>&1; tail -1 $OUT/fuzz_w64.txt
timeout -k 10 500 python tools/fuzz_onchip.py 80 24000 > $OUT/fuzz_onchip.txt 2>&1; tail -1 $OUT/fuzz_onchip.txt
timeout -k 10 500 python tools/fuzz_dense.py > $OUT/fuzz_dense.txt 2>&1; tail -2 $OUT/fuzz_dense.txt
