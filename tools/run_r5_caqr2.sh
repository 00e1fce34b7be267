#!/bin/bash
# round 5: K3 stage 1 -- branch-free panel step + wave priorities by role + the wide apply held behind the urgent level-0 apply when fewer than
# QRK_CAQR_HOLD trailing columns are left (0 = never, default = always), against the build before (old)
OUT=gpurun_out/r5caqr2
mkdir -p $OUT
run() { timeout -k 10 200 python tools/caqr_probe.py 2>&1 | grep factorize | tail -2 | tr '\n' ' '; timeout -k 10 200 python tools/caqr_probe.py 40000 512 2>&1 | grep factorize | tail -1; }
for pass in 1 2; do
  echo "== old"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_old.so run
  for h in 0 800 1200 1600 100000; do echo "== new, hold below $h columns"; QRKIT_AMD_LIB=$PWD/tools/abl/libqrk_new.so QRK_CAQR_HOLD=$h run; done
done > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
