#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace: per (kernel, grid) time per step.  usage: tools/summarize_trace.py <dir> [steps]"""
import collections, csv, glob, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:44]
    wg = int(r["Workgroup_Size_X"])
    key = (n, int(r["Grid_Size_X"]) // wg, r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values()) / steps / 1e3
print(f"total kernel time/step: {tot:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    v2 = sorted(v)
    print(f"{k[0]:44s} wg=({k[1]},{k[2]},{k[3]}) n/step={len(v)/steps:5.1f} med={v2[len(v2)//2]:8.1f}us tot/step={sum(v)/steps/1e3:6.3f}ms")
