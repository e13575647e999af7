#!/usr/bin/env python3
"""Build profiles/<tag>_configs_counters.json and profiles/<tag>_rocprofv3_kernel_stats_configs.csv from the
configs passes of tools/profile_sq.sh (gpurun_out/prof_sq_<tag>/kt_configs, cfg_fetch, cfg_write, cfg_sq).

usage: python tools/configs_summary.py gpurun_out/prof_sq_r02 r02"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short

COMMENT = ("tools/bench_configs.py (configs 2-4 of BASELINE.json + skew cases) under rocprofv3: separate --pmc passes for "
           "FETCH_SIZE (KiB; x2 for wide streaming reads on gfx950), WRITE_SIZE (KiB) and SQ counters, per-launch averages "
           "over ALL launches of a kernel in the script (several workloads share a kernel name); avg_us from the "
           "--kernel-trace --stats pass. Made by tools/profile_sq.sh + tools/configs_summary.py.")


def main():
    root, tag = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for sub in ("cfg_fetch", "cfg_write", "cfg_sq"):
        for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                c = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                c[0] += float(r["Counter_Value"])
                c[1] += 1
    avg_us = {}
    stats = glob.glob(os.path.join(root, "kt_configs", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], f"profiles/{tag}_rocprofv3_kernel_stats_configs.csv")
        for r in csv.DictReader(open(stats[0])):
            k = short(r["Name"])
            t, n = avg_us.get(k, (0.0, 0))
            avg_us[k] = (t + float(r["TotalDurationNs"]) / 1e3, n + int(r["Calls"]))
    out = {}
    for k, v in sorted(acc.items()):
        e = {"launches": max(c[1] for c in v.values())}
        e.update({n: round(c[0] / c[1], 1) for n, c in sorted(v.items())})
        if k in avg_us and avg_us[k][1]:
            e["avg_us"] = round(avg_us[k][0] / avg_us[k][1], 2)
        if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
            e["wait_any_frac_of_wave_cycles"] = round(e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3)
        out[k] = e
    json.dump({"_comment": COMMENT, "kernels": out}, open(f"profiles/{tag}_configs_counters.json", "w"), indent=1)
    print("wrote", len(out), "kernels")


if __name__ == "__main__":
    main()
