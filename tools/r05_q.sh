#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pyramid.py -q -m gpu -x -k "pyramid" 2>&1 | tail -25 | tee gpurun_out/r05q_tests.txt
