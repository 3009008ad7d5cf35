#!/bin/bash
# first GPU session: smoke, parity tests, bench line, kernel sweep, rocprof kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
( timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
( timeout 300 python bench.py --steps 30 --warmup 5 ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -2 gpurun_out/bench.log
( timeout 300 python tools/level_bench.py --rpc 0,8,16,32,64 --generic --rounds 3 ) > gpurun_out/level_bench.log 2>&1; echo "level_bench rc=$?"
cat gpurun_out/level_bench.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
ls -R gpurun_out/prof | head -20
