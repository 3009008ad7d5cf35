#!/bin/bash
# round 3, call 1: aligned detail pitch A/B on the real pyramid kernel (config 2) + baselines of the shapes round 3 works on
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r03a_rowalign.txt
: > $O
for a in 1 8 16 32 64 128 256 1; do
  echo "MIFWT_ROW_ALIGN=$a" >> $O
  MIFWT_ROW_ALIGN=$a timeout 200 python tools/pyr_time.py >> $O 2>&1
done
cat $O
timeout 200 python tools/inv2d_time.py > gpurun_out/r03a_inv2d.txt 2>&1; cat gpurun_out/r03a_inv2d.txt
timeout 300 python tools/r03_refshapes.py > gpurun_out/r03a_refshapes.txt 2>&1; cat gpurun_out/r03a_refshapes.txt
