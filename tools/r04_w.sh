#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_walk3.py -x -q 2>&1 | tail -5 | tee gpurun_out/r04w16_tests.txt
timeout 300 python tools/walk3_st8.py 2>&1 | tee gpurun_out/r04w16_walk3_st8.txt
