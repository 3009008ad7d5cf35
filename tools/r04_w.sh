#!/bin/bash
export TMPDIR=/tmp
timeout 300 python tools/walk3_time2.py 2>&1 | tee gpurun_out/r04w14_walk3_arith.txt
