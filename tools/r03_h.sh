#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for wl in wavedec3_db5_L3_32x100x100x100_f32_periodic wavedec2_db5_L5_32x1000x1000_f32_periodic; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$wl; rm -rf $OUT; mkdir -p $OUT
  ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline ) > $OUT/log.txt 2>&1
  echo "== $wl"; tail -1 $OUT/log.txt | cut -c1-400
  python - <<PY
import csv, glob, collections
t = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    agg[(r["Kernel_Name"][:90], r["Grid_Size_X"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v = sorted(v); print(f"{k[0]:90s} grid {k[1]:>9s} wg {k[2]:>4s} calls {len(v):4d} median {v[len(v)//2]/1e3:8.1f} us total {sum(v)/1e6:8.2f} ms")
PY
done
