#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05l_prio.txt; : > $O
timeout 900 python -W ignore tools/pyr_ab2.py 0,1,2,3,32,33,17 15 2>&1 | grep -v amdgpu | tee -a $O
