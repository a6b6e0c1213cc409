#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ counters (tools/r02_refresh.sh: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS) per kernel.
usage: pmc_sq_summary.py <counter_collection.csv> [clock_GHz]
mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES per SIMD (1024 SIMDs) / (kernel duration * clock); the other columns are fractions of
SQ_WAVE_CYCLES (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)."""
import collections, csv, sys

ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.1
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    a = acc[n][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        d = dur[n]
        d[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[1] += 1
print(f"{'kernel':36s} launches  us/launch  mfma_busy%  active%  wait_any%  wait_inst%  lds_conflict%")
for n, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]][0]):
    if not dur[n][1] or dur[n][0] / dur[n][1] < 5000:
        continue
    wc = c["SQ_WAVE_CYCLES"][0] / c["SQ_WAVE_CYCLES"][1]
    us = dur[n][0] / dur[n][1] / 1000
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / max(c["SQ_VALU_MFMA_BUSY_CYCLES"][1], 1)
    f = lambda k: c[k][0] / max(c[k][1], 1) / wc * 100
    print(f"{n[:36]:36s} {dur[n][1]:8d} {us:10.1f} {mf / 1024 / (us * ghz * 1000) * 100:10.1f} {f('SQ_ACTIVE_INST_ANY'):8.1f} "
          f"{f('SQ_WAIT_ANY'):10.1f} {f('SQ_WAIT_INST_ANY'):11.1f} {f('SQ_LDS_BANK_CONFLICT'):14.1f}")
