#!/bin/bash
# round 4, third GPU call: dense planes again (the default), 8-byte stores with the non-temporal policy, bench line
export TMPDIR=/tmp
O=gpurun_out/r04c_st8_policy.txt; : > $O
for rep in 1 2; do
for cfg in "libmifwt_r3.so 1 0" "libmifwt.so 1 0" "libmifwt_st8nt.so 1 0" "libmifwt.so 16 0" "libmifwt.so 16 512"; do
  set -- $cfg
  MIFWT_LIB=$1 MIFWT_PYRAMID_ROW_ALIGN=$2 timeout 200 python -W ignore tools/pyr_ab.py $3 2>&1 | grep -v amdgpu | tail -1 >> $O
done; done
cat $O
timeout 600 python bench.py > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err; head -c 400 gpurun_out/r04c_bench.json
