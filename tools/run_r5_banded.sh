#!/bin/bash
# round 5: the banded panel's reflector loop with the next pivot column updated first and its head run at once (new) against the build before
OUT=gpurun_out/r5banded
mkdir -p $OUT
for v in old new old new; do echo "== $v"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/strips_probe.py 1024 2>&1 | grep strips; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_$v.so timeout -k 10 200 python tools/banded_probe.py 2>&1 | grep -i "ms" | tail -3; done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout -k 10 900 python -m pytest tests/test_banded_gpu.py tests/test_banded_strips_gpu.py -q -m gpu -x 2>&1 | tail -3 > $OUT/tests.txt; cat $OUT/tests.txt
