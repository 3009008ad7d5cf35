#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03j_pitch_x_policy.txt; : > $O
for lib in libmifwt.so libmifwt_st17.so; do for a in 1 16 128 256; do
  echo "== MIFWT_LIB=$lib MIFWT_ROW_ALIGN=$a" >> $O
  MIFWT_LIB=$lib MIFWT_ROW_ALIGN=$a timeout 200 python tools/pyr_time.py 2>&1 | grep -v amdgpu >> $O
done; done
cat $O
timeout 200 python tools/host_bound.py 2>&1 | grep -v amdgpu | tail -3
