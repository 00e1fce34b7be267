#!/bin/bash
# Timing experiments on the block update of the banded chain (wrong results by construction): see QRK_BB_ABL in banded.hip
for a in ${@:-prof abl1 abl2 abl4 abl3}; do
  echo "== $a"
  QRK_BB_PROF=1 QRKIT_AMD_LIB=$PWD/build/libqrkit_amd_bb$a.so timeout 300 python tools/banded_probe.py 256 2>&1 | grep -E "per panel|factorize" | sed -e 's/.*QR C+D store, larft.: //' -e 's/analyzePattern.*factorize/factorize/'
done
