#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_gpu.log
( timeout 600 python tools/bench_more.py ) 2>/dev/null | tee gpurun_out/bench_more.log | cut -c1-230
