#!/bin/bash
# round 5, K1 experiments (one gpurun call): co-issue probe, A/B of the bdqr_pair4 variants in tools/abl, stamped timelines
OUT=gpurun_out/r5k1a
mkdir -p $OUT
timeout -k 10 120 build/ubench_coissue > $OUT/coissue.txt 2>&1; echo "coissue rc=$?"
QRK_AB_HASH=1 timeout -k 10 400 python tools/ab.py run 10000 > $OUT/ab_10000.txt 2>&1; tail -14 $OUT/ab_10000.txt
timeout -k 10 300 python tools/ab.py run 8192 > $OUT/ab_8192.txt 2>&1; tail -13 $OUT/ab_8192.txt
timeout -k 10 300 python tools/ab.py run 100000 > $OUT/ab_100000.txt 2>&1; tail -13 $OUT/ab_100000.txt
for t in _base _prio8 _prio9; do QRK_P4_TAG=$t timeout -k 10 120 python tools/p4_stamps.py 10000 > $OUT/stamps$t.txt 2>&1; done
head -3 $OUT/stamps_prio8.txt
