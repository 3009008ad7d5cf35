#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_canaries.py -q -m gpu 2>&1 | tail -40 | tee gpurun_out/r05o_canaries.txt
