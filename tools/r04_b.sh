#!/bin/bash
# round 4, second GPU call: parity of both store paths of kernel 16, same-run A/B (round-3 library, 8-byte / 16-byte stores, store policies),
# the new kernel-23 parity tests, a first bench line with rotating outputs + secondary workloads
export TMPDIR=/tmp
O=gpurun_out/r04b_st16_ab.txt; : > $O
timeout 1200 python -m pytest tests/test_gpu_pyramid.py -x -q -m gpu 2>&1 | tail -15 | tee -a $O
for rep in 1 2; do
for cfg in "libmifwt_r3.so 1 0" "libmifwt.so 1 0" "libmifwt.so 16 512" "libmifwt.so 16 0" "libmifwt_st16def.so 16 0" "libmifwt_st16nt.so 16 0"; do
  set -- $cfg
  MIFWT_LIB=$1 MIFWT_PYRAMID_ROW_ALIGN=$2 timeout 200 python -W ignore tools/pyr_ab.py $3 2>&1 | grep -v amdgpu | tail -1 >> $O
done; done
cat $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_slice_reconstruction or batch_and_single or mfma" 2>&1 | tail -15 | tee gpurun_out/r04b_k23_tests.txt
timeout 600 python bench.py > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err; tail -c 3000 gpurun_out/r04b_bench.json
