#!/bin/bash
# round 5: the latency form of bdqr_pair4 (step<.., LAT>) against the throughput form, by batch size; parity with the form forced on
OUT=gpurun_out/r5lat
mkdir -p $OUT
LIB=$PWD/qrkit_amd/lib/libqrkit_amd.so
for B in 2 626 1250 2500 5000 8192 10000 12000 20000 100000; do
  for rep in 1 2; do
    for L in 0 1; do
      r=$(QRK_P4_OWN=$L QRK_AB_HASH=1 QRKIT_AMD_LIB=$LIB timeout -k 10 100 python tools/ab.py one $B 2>/dev/null | tr '\n' ' ')
      echo "B=$B LAT=$L $r"
    done
  done
done > $OUT/lat_ab.txt 2>&1
cat $OUT/lat_ab.txt
QRK_P4_OWN=1 timeout -k 10 600 python -m pytest tests/test_pair_generations_gpu.py tests/test_ties_gpu.py tests/test_margins_gpu.py tests/test_bd_gpu.py tests/test_golden_gpu.py -q -m gpu -x 2>&1 | tail -2 > $OUT/tests_lat1.txt; cat $OUT/tests_lat1.txt
