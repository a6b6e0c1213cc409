#!/usr/bin/env python3
"""dev: timeline of the LAST replayed iteration in a rocprofv3 kernel_trace.csv: per queue, kernels in start order with start offset,
duration and the gap to the previous kernel of the same queue; plus the chip-level union of busy time."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    r["n"] = re.sub(r"^void ", "", n).split("(")[0][:44]
rows.sort(key=lambda r: r["s"])
# iterations end with the second adam_flat_kernel (D's)
adam = [i for i, r in enumerate(rows) if r["n"].startswith("adam_flat")]
last_end = adam[-1]
prev_end = adam[-3]
it = rows[prev_end + 1:last_end + 1]
t0 = it[0]["s"]
print(f"iteration: {len(it)} launches, {(it[-1]['e'] - t0) / 1e3:.1f} us")
queues = sorted(set(r["Queue_Id"] for r in it))
for q in queues:
    qs = [r for r in it if r["Queue_Id"] == q]
    busy = sum(r["e"] - r["s"] for r in qs)
    print(f"== queue {q}: {len(qs)} launches, busy {busy / 1e3:.1f} us, span {(qs[0]['s'] - t0) / 1e3:.1f} .. {(qs[-1]['e'] - t0) / 1e3:.1f}")
    if len(sys.argv) > 2:
        pe = qs[0]["s"]
        for r in qs:
            print(f"   {(r['s'] - t0) / 1e3:8.1f} +{(r['e'] - r['s']) / 1e3:6.1f}  gap {(r['s'] - pe) / 1e3:6.1f}  {r['n']}")
            pe = r["e"]
# union busy
ev = sorted([(r["s"], 1) for r in it] + [(r["e"], -1) for r in it])
lvl = 0; last = None; busy = 0; both = 0
for t, d in ev:
    if lvl > 0:
        busy += t - last
    if lvl > 1:
        both += t - last
    lvl += d; last = t
print(f"chip busy (>=1 kernel) {busy / 1e3:.1f} us, >=2 kernels {both / 1e3:.1f} us, idle {(it[-1]['e'] - t0 - busy) / 1e3:.1f} us")
