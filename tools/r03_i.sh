#!/bin/bash
export TMPDIR=/tmp
timeout 100 ./tools/wbench 2>&1 | tail -9 > gpurun_out/r03i_wbench_rows.txt; cat gpurun_out/r03i_wbench_rows.txt
timeout 200 python tools/inv2d_time.py 2>&1 | grep -v amdgpu | head -3
( timeout 300 python bench.py --workload waverec2_db4_L3_64x1024x1024_f32 --no-cpu-baseline --steps 200 ) 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench waverec2', d['ms_per_step'], d['spinup_steps'], d['roofline']['avg_launch_ms'])"
( timeout 300 python bench.py --workload waverec2_db4_L3_64x1024x1024_f32 --no-cpu-baseline --steps 200 ) 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench waverec2', d['ms_per_step'], d['spinup_steps'], d['roofline']['avg_launch_ms'])"
timeout 300 python tools/small_time.py 2>&1 | grep -v amdgpu | head -8
