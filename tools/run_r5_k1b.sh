#!/bin/bash
# round 5, K1 experiments, second batch: instruction costs, seeds' accuracy, LDS / VALU diet variants of bdqr_pair4
OUT=gpurun_out/r5k1b
mkdir -p $OUT
timeout -k 10 200 build/ubench8 > $OUT/ubench8.txt 2>&1; echo "ubench8 rc=$?"
timeout -k 10 100 build/ubench3 > $OUT/ubench3.txt 2>&1; cat $OUT/ubench3.txt
QRK_AB_HASH=1 timeout -k 10 400 python tools/ab.py run 10000 > $OUT/ab_10000.txt 2>&1; tail -12 $OUT/ab_10000.txt
timeout -k 10 300 python tools/ab.py run 100000 > $OUT/ab_100000.txt 2>&1; tail -11 $OUT/ab_100000.txt
timeout -k 10 300 python tools/ab.py run 1250 > $OUT/ab_1250.txt 2>&1; tail -11 $OUT/ab_1250.txt
