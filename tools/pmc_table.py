"""Per-kernel averages of a rocprofv3 --pmc --kernel-trace csv pair + derived clock / MFMA utilisation.
usage: python tools/pmc_table.py <dir>   (finds *_counter_collection.csv and *_kernel_trace.csv below it)"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
cc = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in cc:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "pm" not in name:
            continue
        key = name[:70] + "  grid=" + r.get("Grid_Size", "?")
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur:
            d[key]["_dur_s"].append(dur[r["Dispatch_Id"]])
for k, v in d.items():
    avg = {c: sum(x) / len(x) for c, x in v.items()}
    line = k + "\n   " + "  ".join(f"{c}={a:.4g}" for c, a in sorted(avg.items()))
    t = avg.get("_dur_s")
    if t and "GRBM_GUI_ACTIVE" in avg:
        clk = avg["GRBM_GUI_ACTIVE"] / 8 / t
        line += f"\n   clock ~ {clk / 1e9:.3f} GHz"
        if "SQ_LDS_IDX_ACTIVE" in avg:  # (LdsUtil of the profiler's derived metrics: LDS cycles in use / (CUs x cycles))
            line += f"  LDS in use {avg['SQ_LDS_IDX_ACTIVE'] / (t * clk * 256):.3f} of (CUs x cycles)"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            line += f"  MFMA pipe busy {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (t * clk * 1024):.3f} of (SIMDs x cycles)"
    print(line)
