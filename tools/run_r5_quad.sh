#!/bin/bash
# round 5: bdqr_quad (four tiles of up to 16 rows per wave) against bdqr_small's 16-lane groups (QRK_QUAD=0), at 5 and 6 waves per SIMD; parity
OUT=gpurun_out/r5quad
mkdir -p $OUT
for v in small quad5 quad6 small quad5 quad6; do
  echo "== $v"
  if [ $v = small ]; then QRK_QUAD=0 timeout -k 10 200 python tools/quad_probe.py 2>&1 | grep " x "
  else QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/quad_probe.py 2>&1 | grep " x "; fi
done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout -k 10 600 python -m pytest tests/test_quad_gpu.py tests/test_small_tiles_gpu.py tests/test_golden_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py tests/test_bd_gpu.py tests/test_lm_gpu.py -q -m gpu -x 2>&1 | tail -4 > $OUT/tests.txt; cat $OUT/tests.txt
