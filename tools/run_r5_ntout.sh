#!/bin/bash
# round 5: non-temporal stores for the coalesced output sweeps of the LDS-staged small-tile kernels (bdqr_thin, bdqr_small): the library
# built with -DQRK_NT_OUT=1 (tools/abl/libqrk_ntout.so) against the shipped one, two interleaved passes; and what the HBM sustains for
# the kernels' read : write mixes (tools/ubench_stream_mix.hip)
OUT=gpurun_out/r5ntout
mkdir -p $OUT
SH=${1:-7x2,9x2,4x4,3x3,16x2,12x1}
for pass in 1 2; do
  echo "== plain stores"; timeout -k 10 200 python tools/small_probe_big.py $SH 2>&1 | grep " B="
  echo "== non-temporal stores"; QRKIT_AMD_LIB=tools/abl/libqrk_ntout.so timeout -k 10 200 python tools/small_probe_big.py $SH 2>&1 | grep " B="
done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
