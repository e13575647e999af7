#!/usr/bin/env python3
"""Build the profiles/ artefacts from gpurun_out/prof_<tag>/ (made by tools/profile_round.sh on the GPU box).

usage: python tools/pmc_traffic.py gpurun_out/prof_r01 r01
writes profiles/<tag>_rocprofv3_kernel_stats_n1e6.csv / _n1e7.csv   (rocprofv3 --kernel-trace --stats, verbatim)
       profiles/<tag>_pmc_fetch_write_n1e6.json / _n1e7.json        (per-kernel per-launch FETCH_SIZE / WRITE_SIZE)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha
from captured_sha import captured_sha  # noqa: E402

COMMENT = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) of `python3 bench.py --n {n} --steps 5 "
           "--warmup 2 --no-cpu-baseline --no-configs --extra-n 0`, per-launch averages in KiB as reported. Per MI355X_MICROARCH.md §HBM: "
           "FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide coalesced streaming reads (so hbm_read_bytes ~= "
           "2*FETCH_SIZE*1024 for those; other access widths uncalibrated; calibrated here on extrema_partial_kernel, "
           "which streams 16 B/leaf); WRITE_SIZE is exact for streaming stores. Infinity-Cache hits are counted, so "
           "these are L2-miss bytes, not HBM bytes.")


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


def main():
    root, tag = sys.argv[1], sys.argv[2]
    os.makedirs("profiles", exist_ok=True)
    for suffix, n in (("", 1_000_000), ("_1e7", 10_000_000)):
        label = "n1e6" if n == 1_000_000 else "n1e7"
        stats = glob.glob(os.path.join(root, "kt" + suffix, "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            shutil.copy(stats[0], f"profiles/{tag}_rocprofv3_kernel_stats_{label}.csv")
        fetch = per_kernel(os.path.join(root, "fetch" + suffix), "FETCH_SIZE")
        write = per_kernel(os.path.join(root, "write" + suffix), "WRITE_SIZE")
        if not fetch and not write:
            continue
        kernels = {}
        for name in sorted(set(fetch) | set(write)):
            f, w = fetch.get(name, [0.0, 0]), write.get(name, [0.0, 0])
            kernels[name] = {"launches": max(f[1], w[1]),
                             "FETCH_SIZE_KiB_avg": round(f[0] / f[1], 2) if f[1] else None,
                             "WRITE_SIZE_KiB_avg": round(w[0] / w[1], 2) if w[1] else None}
        # csrc_sha: the state of the kernel sources these counters belong to (run this script BEFORE touching csrc/ again);
        # bench.py quotes the figures only while the hash still matches
        json.dump({"_comment": COMMENT.format(n=n), "csrc_sha": captured_sha(root), "kernels": kernels},
                  open(f"profiles/{tag}_pmc_fetch_write_{label}.json", "w"), indent=1)
        print("wrote", label, len(kernels), "kernels")


if __name__ == "__main__":
    main()
