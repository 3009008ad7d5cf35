#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "tile" ) 2>&1 | tail -2
for shp in 64,261,261 64,515,515 64,1024,1024; do
  ( timeout 300 python tools/level_bench.py --shape $shp --tile 2,1 --tr 0 --rounds 5 --iters 20 ) 2>/dev/null | cut -c1-160
done
python tools/copy_floor.py 2>/dev/null
