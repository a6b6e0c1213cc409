#!/usr/bin/env python3
"""dev: per-kernel time per step of two rocprofv3 kernel_stats.csv files side by side (steps = launches of adam_flat_kernel / 2)."""
import csv, sys, re


def load(path):
    rows = list(csv.DictReader(open(path)))
    steps = next(int(r["Calls"]) for r in rows if "adam_flat_kernel" in r["Name"]) / 2
    out = {}
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        out[name] = (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e3)
    return out, steps


a, sa = load(sys.argv[1])
b, sb = load(sys.argv[2])
names = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0))[1] + b.get(n, (0, 0))[1]))
ta = tb = 0
la = lb = 0
print(f"{'kernel':60s} {'A calls':>8s} {'A us':>9s} {'B calls':>8s} {'B us':>9s} {'diff':>8s}")
for n in names:
    ca, ua = a.get(n, (0, 0))
    cb, ub = b.get(n, (0, 0))
    ta += ua; tb += ub; la += ca; lb += cb
    if max(ua, ub) >= 5:
        print(f"{n[:60]:60s} {ca:8.1f} {ua:9.1f} {cb:8.1f} {ub:9.1f} {ub - ua:8.1f}")
print(f"{'TOTAL':60s} {la:8.1f} {ta:9.1f} {lb:8.1f} {tb:9.1f} {tb - ta:8.1f}")
