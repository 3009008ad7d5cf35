#!/bin/bash
# rocprofv3 kernel trace of one command, summarised per (kernel, grid size): calls, average / median duration, and the gaps between
# consecutive kernels.  usage: bash tools/ktrace.sh <tag> <command ...>   -> gpurun_out/<tag>_ktrace.txt
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$ROOT/gpurun_out/kt_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- "$@" ) > $OUT/log.txt 2>&1; echo "kernel-trace rc=$?"
cd $ROOT
python - <<PY
import csv, glob, collections
rows = []
for p in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list); gaps = collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    key = (r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))
    by[key].append(e - s)
    if prev_end is not None: gaps[key].append(s - prev_end)
    prev_end = e
with open("gpurun_out/${TAG}_ktrace.txt", "w") as f:
    for key, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        sv = sorted(v[len(v) // 3:]); g = sorted(gaps[key][len(gaps[key]) // 3:]) or [0]
        line = "%-90s grid %8s wg %5s calls %5d  avg %8.1f us  median %8.1f  gap before (median) %6.1f us" % (key[0], key[1], key[2], len(v), sum(sv) / len(sv) / 1e3, sv[len(sv) // 2] / 1e3, g[len(g) // 2] / 1e3)
        print(line); f.write(line + "\n")
PY
