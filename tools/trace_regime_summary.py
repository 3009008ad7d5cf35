"""Summary of a rocprofv3 --kernel-trace CSV of tools/trace_regime.py: launches of the headline kernel AFTER the marker fill, as one CSV
row set (calls, average, median, min, max, sigma in ns) + the fraction of the 8 TB/s peak the AVERAGE corresponds to.
usage: trace_regime_summary.py kernel_trace.csv out.csv regime"""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cut = max((i for i, r in enumerate(rows) if "fill" in r["Kernel_Name"].lower() or "FillFunctor" in r["Kernel_Name"]), default=-1)
# the marker = the LAST fill kernel before the traced loop: the last one overall precedes only the traced launches
sel = [r for r in rows[cut + 1:] if "dwt2_fwd_pyr_kernel" in r["Kernel_Name"]]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
BYTES = 542832128
avg = sum(d) / len(d)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Regime", "Calls", "AverageNs", "MedianNs", "MinNs", "MaxNs", "StdDevNs", "AlgorithmicBytes", "AverageFracOf8TBps", "MedianFracOf8TBps"])
    w.writerow([sel[0]["Kernel_Name"][:100], sys.argv[3], len(d), round(avg, 1), statistics.median(d), min(d), max(d), round(statistics.pstdev(d), 1), BYTES,
                round(BYTES / avg / 8000.0, 4), round(BYTES / statistics.median(d) / 8000.0, 4)])
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(sel, sel[1:])]
    w.writerow(["# gap between consecutive launches (ns): average", round(sum(gaps) / len(gaps), 1), "median", statistics.median(gaps)])
    span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / len(sel)
    w.writerow(["# wall time per call over the traced loop (ns)", round(span, 1), "frac of 8 TB/s on the compulsory bytes", round(BYTES / span / 8000.0, 4)])
print(open(sys.argv[2]).read())
