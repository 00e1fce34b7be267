#!/bin/bash
# Direct level-2 path vs the two-stage form of the pivoted dense QR over a few shapes (GPU box); picks the crossover of the plan's default.
for shape in "5120 384" "10000 256" "40000 128" "40000 200" "10000 500" "40000 500" "20000 1000" "8000 2000" "4000 1000" "2000 500"; do
  for ts in 0 1; do
    QRK_DENSE_TWO_STAGE=$ts python3 tools/caqr_probe.py $shape 2>&1 | grep factorize | tail -1
  done
done
