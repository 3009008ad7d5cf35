#!/bin/bash
export TMPDIR=/tmp
for shp in 64,261,261 64,515,515; do
  ( timeout 300 python tools/level_bench.py --shape $shp --tile 2 --rpc 4,6,8 --depth 1,2,3 --rounds 3 --iters 20 ) 2>/dev/null | cut -c1-160
done
