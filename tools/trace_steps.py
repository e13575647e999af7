"""Per-kernel average duration and the gap behind every launch over the last 100 steps of a rocprofv3 --kernel-trace CSV
(a step = from one extrema_partial_kernel launch to the next).  usage: python tools/trace_steps.py <r_kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("extrema_partial")]
steps = min(100, len(idx) - 2)
per, gaps = collections.OrderedDict(), collections.OrderedDict()
for s in range(len(idx) - 1 - steps, len(idx) - 1):
    i0, i1 = idx[s], idx[s + 1]
    for k in range(i0, i1):
        key = (k - i0, names[k])
        per[key] = per.get(key, 0) + int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])
        gaps[key] = gaps.get(key, 0) + int(rows[k + 1]["Start_Timestamp"]) - int(rows[k]["End_Timestamp"])
tot = 0
for k, v in per.items():
    print(f"{k[0]:2d} {k[1]:32s} {v / steps / 1000:7.2f} us   gap after {gaps[k] / steps / 1000:6.2f} us")
    tot += v + gaps[k]
print(f"period {tot / steps / 1000:.2f} us; kernel sum {sum(per.values()) / steps / 1000:.2f} us over {steps} steps")
