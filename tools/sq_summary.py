#!/usr/bin/env python3
"""Build profiles/<tag>_sq_counters_n1e6.json from the SQ counter passes of tools/profile_sq.sh
(gpurun_out/prof_sq_<tag>/sq_a|sq_b|sq_c): per-launch averages per kernel, keyed like bench.py's `kernels`.

usage: python tools/sq_summary.py gpurun_out/prof_sq_r02 r02
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over the waves or
SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Derived: kernel cycles = GRBM_GUI_ACTIVE / 8; busy fraction of a
unit = 4 * SQ_ACTIVE_INST_x / (1024 SIMDs * kernel cycles)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha, pmc_key
from captured_sha import captured_sha  # noqa: E402


def main():
    root, tag = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "sq_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = acc[pmc_key(r["Kernel_Name"])][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    out = {}
    for k, v in sorted(acc.items()):
        e = {n: round(c[0] / c[1], 1) for n, c in sorted(v.items())}
        e["launches"] = max(c[1] for c in v.values())
        cyc = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc > 0:
            e["kernel_cycles"] = round(cyc, 1)
            if "SQ_ACTIVE_INST_VALU" in e:
                e["valu_busy_frac"] = round(4 * e["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 3)
            if "SQ_ACTIVE_INST_SCA" in e:
                e["salu_busy_frac"] = round(4 * e["SQ_ACTIVE_INST_SCA"] / (1024 * cyc), 3)
            if "SQ_WAVE_CYCLES" in e and "SQ_WAIT_ANY" in e:
                e["wave_parked_frac"] = round(e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"], 3)
                e["wave_issue_stall_frac"] = round(e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3)
        if "SQ_WAVES" in e and e["SQ_WAVES"] > 0:
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if n in e:
                    e[n.lower().replace("sq_insts_", "") + "_per_wave"] = round(e[n] / e["SQ_WAVES"], 1)
        out[k] = e
    os.makedirs("profiles", exist_ok=True)
    json.dump({"_comment": "rocprofv3 --pmc SQ counter passes (three separate passes, no tracing flags) of `python3 bench.py "
                           "--no-cpu-baseline --no-configs --extra-n 0 --steps 3 --warmup 1` (1e6 leaves), per-launch averages; made by "
                           "tools/profile_sq.sh + tools/sq_summary.py", "csrc_sha": captured_sha(root), "kernels": out},
              open(f"profiles/{tag}_sq_counters_n1e6.json", "w"), indent=1)
    print("wrote", len(out), "kernels")


if __name__ == "__main__":
    main()
