#!/usr/bin/env python
"""Condense rocprofv3 output under gpurun_out/pmc (made by tools/gpu_pmc.sh) into small committed summaries
under profiles/:  <tag>_kernel_stats.csv (per-kernel time, --kernel-trace --stats) and <tag>_pmc_level1.json
(PMC counters of the dominant kernel, averaged per dispatch, with the gfx950 FETCH_SIZE x2 correction of
/opt/skills/guides/MI355X_MICROARCH.md §HBM applied)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/pmc"
os.makedirs("profiles", exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    for key in ("idwt1_long_kernel", "idwt1_tail_kernel", "idwt2_small_kernel", "idwt3_tile_kernel", "dwt2_fwd_small_kernel", "dwt2_fwd_pyr_kernel", "dwt2_fwd_roll_kernel", "dwt2_fwd_pair_kernel", "dwt2_fwd_tile_kernel", "dwt2_fwd_stream_kernel", "dwt2_inv_stream_kernel", "outer_fwd_kernel", "outer_inv_kernel", "inner_fwd_kernel", "inner_inv_kernel", "swt_kernel", "axis_adj_kernel", "axis_fwd_kernel", "axis_inv_kernel", "dwt3_", "dwt1_"):
        if key in name:
            i = name.index(key)
            j = name.find("(", i)
            return "mifwt::" + name[i:j if j > 0 else None]
    return name[:60] + ("..." if len(name) > 60 else "")


stats = glob.glob(os.path.join(src, "kt", "*kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    print("wrote", f"profiles/{tag}_kernel_stats.csv")
trace = glob.glob(os.path.join(src, "kt", "*kernel_trace.csv"))
if trace:
    # per-level durations of the fused kernel: group by grid size
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        if "dwt2_fwd_" in r["Kernel_Name"]:
            agg[(short(r["Kernel_Name"]), r["Grid_Size_X"], r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(f"profiles/{tag}_kernel_trace_by_level.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Grid_Size", "VGPR_Count", "LDS_Block_Size", "Calls", "AverageNs", "MinNs", "MaxNs"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([*k, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
    print("wrote", f"profiles/{tag}_kernel_trace_by_level.csv")

pmc = {}
durs = []
for p in sorted(glob.glob(os.path.join(src, "p*", "*_counter_collection.csv"))):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "dwt2_fwd_" not in r["Kernel_Name"]:
            continue
        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        pmc.setdefault("_kernel", short(r["Kernel_Name"]))
        pmc.setdefault("_grid", r["Grid_Size"])
        pmc.setdefault("_vgpr", r["VGPR_Count"])
    for k, v in per.items():
        pmc[k] = sum(v) / len(v)
if pmc:
    out = {"kernel": pmc.pop("_kernel"), "grid_size": pmc.pop("_grid"), "vgpr_count": pmc.pop("_vgpr"),
           "counters_per_dispatch": {k: round(v, 1) for k, v in sorted(pmc.items())}}
    c = out["counters_per_dispatch"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reports 1/2 of a wide coalesced stream
        out["hbm_read_bytes_corrected"] = int(c["FETCH_SIZE"] * 1024 * 2)
        out["hbm_write_bytes"] = int(c["WRITE_SIZE"] * 1024)
        out["hbm_traffic_bytes"] = out["hbm_read_bytes_corrected"] + out["hbm_write_bytes"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        out["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    json.dump(out, open(f"profiles/{tag}_pmc_level1.json", "w"), indent=1)
    print("wrote", f"profiles/{tag}_pmc_level1.json")
    print(json.dumps(out, indent=1))
