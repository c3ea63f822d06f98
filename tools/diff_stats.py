"""Per-forward kernel table = (stats of N2 forwards - stats of N1 forwards) / (N2 - N1).
usage: python tools/diff_stats.py a_kernel_stats.csv N1 b_kernel_stats.csv N2"""
import csv
import sys


def load(p):
    d = {}
    with open(p) as f:
        for r in csv.DictReader(f):
            d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return d


a, n1, b, n2 = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
rows = []
for k, (cb, tb) in b.items():
    ca, ta = a.get(k, (0, 0.0))
    dc, dt = (cb - ca) / (n2 - n1), (tb - ta) / (n2 - n1)
    if dc > 0.01:
        rows.append((dt, dc, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"# per forward: {tot / 1e6:.3f} ms of kernel time in {sum(r[1] for r in rows):.0f} launches")
for dt, dc, k in rows:
    print(f"{dt / 1e6:8.3f} ms {dc:7.1f} calls {dt / dc / 1e3:8.1f} us  {k[:150]}")
