#!/bin/bash
# round 5: CAQR panel kernel with the reflector's scalars computed beside the dot products (the owners of a column leave |x_tail|^2 and the
# pivot entry when they publish it) against the build before: 40 000 x 2 000 pivoted (two-stage), 100 000 x 512 un-pivoted (stage 1 alone)
OUT=gpurun_out/r5caqr
mkdir -p $OUT
for v in old noprio prio old noprio prio; do echo "== $v"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/caqr_probe.py 2>&1 | grep factorize | tail -2; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/caqr_probe.py 40000 512 2>&1 | grep factorize | tail -1; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout -k 10 900 python -m pytest tests/test_dense_gpu.py tests/test_thin_gpu.py tests/test_angular.py -q -m gpu -x 2>&1 | tail -3 > $OUT/tests.txt; cat $OUT/tests.txt
