"""Average the counters of a rocprofv3 --pmc run per pm:: kernel.
usage: python tools/pmc_summary.py out_dir/prefix_counter_collection.csv"""
import collections
import csv
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "pm" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    line = k + " :: " + "  ".join(f"{c}={sum(x) / len(x):.4g}" for c, x in sorted(v.items()))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v:
        line += "  | MFMA busy / SQ busy = %.3f" % (sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / sum(v["SQ_BUSY_CYCLES"]))
    print(line)
