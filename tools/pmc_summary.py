#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: average counter value per launch.

usage: python tools/pmc_summary.py gpurun_out/<dir> [substring filter]  -> JSON on stdout
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.match(r"(?:void )?(?:ibvh::)?(?:\w+::)*(\w+)(<.*>)?\(", name)
    if not m:
        return name[:60]
    base, targs = m.group(1), m.group(2) or ""
    flags = re.findall(r"\b(true|false)\b", targs)
    return base + ("<" + ",".join(flags) + ">" if flags else "")


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if flt and flt not in k:
                continue
            c = acc[k][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    out = {k: {"launches": max(c[1] for c in v.values()), **{n: c[0] / c[1] for n, c in sorted(v.items())}} for k, v in sorted(acc.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
