#!/usr/bin/env python3
"""profiles/<tag>_workload_counters.json from gpurun_out/prof_wl_<tag>/ (tools/profile_workloads.sh): per workload and kernel
the average launch duration (rocprofv3 --kernel-trace --stats), FETCH_SIZE / WRITE_SIZE per launch and the L2-miss traffic
2 * FETCH + WRITE (MI355X_MICROARCH.md §HBM).  Only kernels of the library (ibvh::) are kept.
usage: python tools/workload_summary.py gpurun_out/prof_wl_r04 r04"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha, pmc_key
from captured_sha import captured_sha  # noqa: E402


def counters(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "ibvh::" in r["Kernel_Name"]:
                a = acc[pmc_key(r["Kernel_Name"])]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return acc


root, tag = sys.argv[1], sys.argv[2]
out = {}
for wdir in sorted(glob.glob(os.path.join(root, "*/"))):
    w = os.path.basename(wdir.rstrip("/"))
    info = None
    try:
        info = json.loads([l for l in open(os.path.join(root, w + ".kt.log")) if l.startswith("{")][-1])
    except Exception:
        pass
    ks = {}
    for f in glob.glob(os.path.join(wdir, "kt", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ibvh::" not in r["Name"]:
                continue
            k = pmc_key(r["Name"])
            e = ks.setdefault(k, {"calls": 0, "total_ns": 0.0})
            e["calls"] += int(r["Calls"])
            e["total_ns"] += float(r["TotalDurationNs"])
    fetch, write = counters(os.path.join(wdir, "fetch"), "FETCH_SIZE"), counters(os.path.join(wdir, "write"), "WRITE_SIZE")
    rows = {}
    for k, e in ks.items():
        f, wv = fetch.get(k), write.get(k)
        row = {"launches": e["calls"], "avg_us": round(e["total_ns"] / e["calls"] / 1e3, 2), "total_ms": round(e["total_ns"] / 1e6, 4)}
        if f and f[1]:
            row["FETCH_SIZE_KiB_avg"] = round(f[0] / f[1], 1)
        if wv and wv[1]:
            row["WRITE_SIZE_KiB_avg"] = round(wv[0] / wv[1], 1)
        if "FETCH_SIZE_KiB_avg" in row and "WRITE_SIZE_KiB_avg" in row:
            row["traffic_MB_per_launch"] = round((2 * row["FETCH_SIZE_KiB_avg"] + row["WRITE_SIZE_KiB_avg"]) * 1024 / 1e6, 2)
        rows[k] = row
    out[w] = {"info": info, "kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["total_ms"]))}
json.dump({"_comment": "one rocprofv3 process per workload and pass (tools/profile_workloads.sh -> tools/profile_workload.py NAME: 2 warm-up + 5 "
                       "timed calls of that workload only); avg_us from --kernel-trace --stats, FETCH / WRITE from separate --pmc passes, "
                       "traffic = 2*FETCH_SIZE + WRITE_SIZE (L2-miss bytes; Infinity-Cache hits included); set-up kernels (generators, "
                       "builds) appear with their own names", "csrc_sha": captured_sha(root), "workloads": out},
          open(f"profiles/{tag}_workload_counters.json", "w"), indent=1)
print("wrote", len(out), "workloads")
