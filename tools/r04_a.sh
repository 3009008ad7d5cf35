#!/bin/bash
# round 4, first GPU call: parity of the 16-byte store path of kernel 16, then same-run A/B against the 8-byte stores and store policies
export TMPDIR=/tmp
O=gpurun_out/r04a_st16_ab.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_pyramid.py -x -q -m gpu 2>&1 | tail -5 | tee -a $O
MIFWT_PYRAMID_ROW_ALIGN=1 timeout 900 python -m pytest tests/test_gpu_pyramid.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O
for rep in 1 2; do
for cfg in "libmifwt.so 16 0" "libmifwt.so 16 512" "libmifwt.so 1 0" "libmifwt_st16def.so 16 0" "libmifwt_st16nt.so 16 0" "libmifwt_st16sc1.so 16 0"; do
  set -- $cfg
  MIFWT_LIB=$1 MIFWT_PYRAMID_ROW_ALIGN=$2 timeout 200 python tools/pyr_ab.py $3 2>&1 | grep -v amdgpu | tail -1 >> $O
done; done
cat $O
