#!/bin/bash
export TMPDIR=/tmp
for st in 20 50 200 1000; do
timeout 300 python bench.py --steps $st --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($st, d['ms_per_step'], d['whole_call']['level_kernel_ms'])"
done
timeout 300 python bench.py --steps 50 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('w50', d['ms_per_step'], d['whole_call']['level_kernel_ms'])"
