#!/bin/bash
# rocprofv3 evidence for ONE bench.py workload: kernel-trace + stats summary, and PMC passes (FETCH_SIZE / WRITE_SIZE each in its own
# pass, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) condensed into profiles-ready files under gpurun_out/:
#   <TAG>_kernel_stats_<WL>.csv   per-kernel time of `python bench.py --workload WL` (rocprofv3 --kernel-trace --stats)
#   <TAG>_pmc_<WL>.json           counters per dispatch of the kernel whose name contains KERNEL (the dominant one)
# usage: TAG=r03 WL=waverec2_db4_L3_64x1024x1024_f32 KERNEL=idwt2_pyr_kernel bash tools/pmc_workload.sh
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmcw_$WL
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --workload $WL --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-secondary"
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD ) > $OUT/kt.log 2>&1; echo "kernel-trace rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  ( timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- $CMD ) > $OUT/p$i.log 2>&1; echo "pmc pass $i rc=$?"
done
cd $ROOT
python - <<PY
import collections, csv, glob, json
out, kern, wl, tag = "$OUT", "$KERNEL", "$WL", "${TAG:-r03}"
kern2 = "${KERNEL2:-}"  # a second kernel of the same level (composed routes: two launches per level): its traffic is added
stats = glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(f"gpurun_out/{tag}_kernel_stats_{wl}.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:12]:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
# per grid size: the same kernel name can serve launches of very different sizes
trace = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)
bygrid = collections.defaultdict(list)
allgrid = collections.defaultdict(list)
if trace:
    for r in csv.DictReader(open(trace[0])):
        g = r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        allgrid[(r["Kernel_Name"][:110], g)].append(dur)
        if kern in r["Kernel_Name"]:
            bygrid[g].append(dur)
if stats and allgrid:
    # (round 4) the per-name averages above mix launches of every grid size: the same summary split by (kernel, grid size), the first
    # quarter of each group (idle-clock spin-up launches) left out of the steady-state column
    top = [r["Name"][:110] for r in rows[:8]]
    with open(f"gpurun_out/{tag}_kernel_stats_{wl}.csv", "a", newline="") as f:
        w = csv.writer(f)
        w.writerow([])
        w.writerow(["# by grid size", "GridSizeX", "Calls", "AverageNs", "MedianNs", "SteadyAverageNs(last 3/4)", "MinNs", "MaxNs"])
        for name in top:
            for (n, g), v in sorted(allgrid.items(), key=lambda kv: -sum(kv[1])):
                if n != name:
                    continue
                sv = v[len(v) // 4:]
                w.writerow([n, g, len(v), round(sum(v) / len(v), 1), sorted(v)[len(v) // 2], round(sum(sv) / len(sv), 1), min(v), max(v)])
pmc, meta = {}, {}
for p in sorted(glob.glob(out + "/p*/**/*_counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(p)) if kern in r["Kernel_Name"]]
    # the dominant launch = the grid of that kernel with the longest launches in the kernel trace (the same kernel serves the other
    # levels; round 4: the 129^3 level of the 3-D synthesis walk has MORE workgroups than the 256^3 level — shorter depth segments —
    # so "the biggest grid" picked the wrong one); without a trace: the biggest grid
    gmax = max((int(r["Grid_Size"]) for r in rows), default=0)
    if bygrid:
        gdom = max(bygrid.items(), key=lambda kv: sorted(kv[1])[len(kv[1]) // 2])[0]
        cand = [int(r["Grid_Size"]) for r in rows]
        # (kernel-trace grid = Grid_Size_X, counter-collection grid = total work-items: same number for 1-D launches)
        if int(gdom) in cand:
            gmax = int(gdom)
    for r in rows:
        if int(r["Grid_Size"]) != gmax:
            continue
        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta.setdefault("kernel", r["Kernel_Name"][:110]); meta.setdefault("grid_size", r["Grid_Size"]); meta.setdefault("vgpr_count", r["VGPR_Count"])
    for k, v in per.items():
        v = v[len(v) // 4:]  # drop the spin-up launches
        # persistent kernels launch the same grid for every level: the dominant launches are the ones with the big counts
        v = [x for x in v if x >= 0.5 * max(v)] if v and max(v) > 0 else v
        pmc[k] = sum(v) / max(1, len(v))
pmc2 = {}
if kern2:
    for p in sorted(glob.glob(out + "/p*/**/*_counter_collection.csv", recursive=True)):
        per = collections.defaultdict(list)
        rows2 = [r for r in csv.DictReader(open(p)) if kern2 in r["Kernel_Name"]]
        gmax = max((int(r["Grid_Size"]) for r in rows2), default=0)
        for r in rows2:
            if int(r["Grid_Size"]) == gmax and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta.setdefault("kernel2", r["Kernel_Name"][:110])
        for k, v in per.items():
            v = v[len(v) // 4:]
            pmc2[k] = sum(v) / max(1, len(v))
res = dict(meta)
res["workload"] = wl
res["counters_per_dispatch"] = {k: round(v, 1) for k, v in sorted(pmc.items())}
res["launch_ns_by_grid"] = {g: {"calls": len(v), "median": sorted(v)[len(v) // 2], "mean_last_half": round(sum(v[len(v) // 2:]) / max(1, len(v) - len(v) // 2), 1)} for g, v in bygrid.items()}
c = res["counters_per_dispatch"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reports 1/2 of a wide coalesced stream (MI355X_MICROARCH.md, HBM section)
    res["hbm_read_bytes_corrected"] = int(c["FETCH_SIZE"] * 1024 * 2)
    res["hbm_write_bytes"] = int(c["WRITE_SIZE"] * 1024)
    res["hbm_traffic_bytes"] = res["hbm_read_bytes_corrected"] + res["hbm_write_bytes"]
    if "FETCH_SIZE" in pmc2 and "WRITE_SIZE" in pmc2:
        res["kernel2_hbm_traffic_bytes"] = int(pmc2["FETCH_SIZE"] * 1024 * 2) + int(pmc2["WRITE_SIZE"] * 1024)
        res["hbm_traffic_bytes"] += res["kernel2_hbm_traffic_bytes"]  # (the level = both launches)
json.dump(res, open(f"gpurun_out/{tag}_pmc_{wl}.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf $OUT
