#!/bin/bash
# round 5: bdqr_w64 with the pivot lane's own norm (QRK_W64_OWN) against the DPP row sum, parity of the new form
OUT=gpurun_out/r5w64
mkdir -p $OUT
for v in own0 own1 own0 own1; do echo "== $v"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/mixed_probe.py 4000 2>&1 | grep -E "uniform (33|40|48|56|64)|mixed"; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_own1.so timeout -k 10 600 python -m pytest tests/test_w64_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py tests/test_onchip_gpu.py -q -m gpu -x 2>&1 | tail -2 > $OUT/tests.txt; cat $OUT/tests.txt
