#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r05j_gpu_suite.txt
timeout 900 python bench.py > gpurun_out/r05j_bench_default.json 2> gpurun_out/r05j_bench_err.txt; tail -3 gpurun_out/r05j_bench_err.txt
