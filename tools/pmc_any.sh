#!/bin/bash
# PMC passes over an arbitrary python snippet; prints per-dispatch averages for kernels whose name contains $KERNEL
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_any
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  ( timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/$SCRIPT ) > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
for kern in "$KERNEL".split(","):  # comma list of kernel-name substrings; the first dispatch of each is dropped (cold)
    print("==", kern)
    for p in sorted(glob.glob("$OUT/p*/*_counter_collection.csv")):
        per = collections.defaultdict(list)
        g = v = l = None
        for r in csv.DictReader(open(p)):
            if kern in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                g = r["Grid_Size"]; v = r["VGPR_Count"]; l = r.get("LDS_Block_Size")
        print(p.split("/")[-2], "grid", g, "vgpr", v, "lds", l, {k: round(sum(x[1:]) / max(1, len(x) - 1)) for k, x in per.items()})
    for p in sorted(glob.glob("$OUT/p1/*_kernel_trace.csv")):
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(p)) if kern in r["Kernel_Name"]]
        print("durations ns", d[:12])
PY
