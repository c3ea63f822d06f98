"""Per-shape A/B of two environment settings with tools/shape_profile.py tables.
usage: python tools/shape_ab.py <table_a.txt> <table_b.txt> [op-prefix]"""
import re
import sys


def load(f, prefix):
    d = {}
    for ln in open(f):
        m = re.match(r"(" + prefix + r" M=\S+ N=\S+ K=\S+ .*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)", ln)
        if m:
            d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    return d


prefix = sys.argv[3] if len(sys.argv) > 3 else "gemm"
a, b = load(sys.argv[1], prefix), load(sys.argv[2], prefix)
ta = tb = 0.0
for k, (n, ms, us) in sorted(a.items(), key=lambda kv: -kv[1][1]):
    if k in b:
        ta += ms
        tb += b[k][1]
        print("%-62s n=%3d  a %7.1f us  b %7.1f us  %+5.0f%%" % (k[:62], n, us, b[k][2], 100 * (b[k][2] / us - 1)))
print("sum %s ms/fwd: a %.2f  b %.2f" % (prefix, ta, tb))
