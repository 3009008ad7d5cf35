#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r05i_calib.txt; : > $O
timeout 1200 python -W ignore tools/pyr_calib.py 2>&1 | grep -v amdgpu | tee -a $O
