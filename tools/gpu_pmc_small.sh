#!/bin/bash
# PMC passes over one analysis level of a chosen shape (default: config-2 level 3)
SHAPE=${SHAPE:-64,261,261}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_small
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/level_bench.py --shape $SHAPE --tile ${TILE:-2} --rounds 1 --iters 4"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  ( timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- $CMD ) > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/p*/*_counter_collection.csv")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "dwt2_fwd" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            g = r["Grid_Size"]; v = r["VGPR_Count"]
    print(p.split("/")[-2], "grid", g, "vgpr", v, {k: round(sum(v)/len(v)) for k, v in per.items()})
for p in sorted(glob.glob("$OUT/p1/*_kernel_trace.csv")):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(p)) if "dwt2_fwd" in r["Kernel_Name"]]
    print("durations ns", d)
PY
