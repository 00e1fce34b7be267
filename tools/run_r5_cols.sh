#!/bin/bash
# round 5: cols_step_kernel (K3 stage 2, the column-parallel pivoted QR) with all loads of a thread in flight in the pivot search and the
# copy of the pivot column, clamped column loads (new) against the build before (old)
OUT=gpurun_out/r5cols
mkdir -p $OUT
run() { timeout -k 10 200 python tools/caqr_probe.py 2>&1 | grep factorize | tail -2 | tr '\n' ' '; timeout -k 10 200 python tools/caqr_probe.py 2000 2000 2>&1 | grep factorize | tail -1 | tr '\n' ' '; timeout -k 10 200 python tools/caqr_probe.py 5120 384 2>&1 | grep factorize | tail -1; }
for v in old new old new; do echo "== $v"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so run; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout -k 10 900 python -m pytest tests/test_dense_gpu.py tests/test_thin_gpu.py tests/test_angular.py tests/test_dense_pers_gpu.py -q -m gpu -x 2>&1 | tail -3 > $OUT/tests.txt; cat $OUT/tests.txt
