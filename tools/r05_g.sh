#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05g_clock.txt; : > $O
for B in 80 65 100 16; do timeout 300 python -W ignore tools/pyr_clock.py $B 0 2>&1 | grep -v amdgpu | tee -a $O; done
