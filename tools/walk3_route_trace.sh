#!/bin/bash
# GPU time per call (sum of the kernels of one level, from a kernel trace — an eager loop of such small calls is host-bound) of the 3-D
# analysis routes: tile mode 1 (bricks / composed) against 4 (depth-walking kernel), 32 volumes.  -> gpurun_out/<tag>_walk3_routes.txt
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r06k}
OUT=$ROOT/gpurun_out/${TAG}_walk3_routes.txt
: > $OUT
export TMPDIR=/tmp
# CFGS="db5:100 db4:53 ..." picks the (wavelet, extent) pairs; BATCH the number of volumes
for c in ${CFGS:-db2:100 db2:51 db2:27 db3:100 db3:52 db3:28 db4:100 db4:53 db4:30 db5:100 db5:54 db5:31}; do
  cfg="${c%%:*} ${c##*:}"
  for tm in 1 4; do
    D=/tmp/w3_$$; rm -rf $D; mkdir -p $D
    ( cd /tmp; timeout 120 rocprofv3 --kernel-trace --output-format csv -d $D -o kt -- python $ROOT/tools/walk3_one.py $cfg $tm ${BATCH:-32} ) > $D/log.txt 2>&1
    python - <<PY >> $OUT
import csv, glob
rows = []
for p in glob.glob("$D/**/*kernel_trace.csv", recursive=True): rows += list(csv.DictReader(open(p)))
rows = [r for r in rows if "mifwt" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = sorted(set(r["Kernel_Name"].split("(")[0][:60] for r in rows))
per = len(rows) // 60 if rows else 0
last = rows[len(rows) // 2:]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / max(1, len(last)) * per / 1e3
print("$cfg batch ${BATCH:-32} tile mode $tm: %.1f us GPU per call (%d launches: %s)" % (tot, per, "; ".join(names)))
PY
  done
done
cat $OUT
