#!/bin/bash
export TMPDIR=/tmp
timeout 300 python tools/walk3_nt.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04x6_walk3_nt.txt
