#!/usr/bin/env python3
"""Turn what one GPU run of
    python bench.py > gpurun_out/final_bench.json; tools/profile_round.sh TAG; tools/profile_sq.sh TAG;
    tools/profile_workloads.sh TAG; tools/sq_rays.sh TAG
left under gpurun_out/ into the committed files of profiles/ (all stamped with the hash of the kernel sources they were
taken at).  usage: python tools/fold_profiles.py r04"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from captured_sha import captured_sha  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
go = os.path.join(ROOT, "gpurun_out")
for script, src in (("pmc_traffic.py", f"prof_{tag}"), ("sq_summary.py", f"prof_sq_{tag}"), ("workload_summary.py", f"prof_wl_{tag}")):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), os.path.join(go, src), tag], check=True)
rays = os.path.join(ROOT, "profiles", f"{tag}_rays_sq_counters.json")
doc = json.load(open(rays)) if os.path.exists(rays) else {}
doc.update({"csrc_sha": captured_sha(os.path.join(go, f"sqr_{tag}")),
            "binned": json.load(open(os.path.join(go, f"sqr_{tag}", "summary_binned.json"))),
            "walker": json.load(open(os.path.join(go, f"sqr_{tag}", "summary_walker.json")))})
json.dump(doc, open(rays, "w"), indent=1)
line = open(os.path.join(go, "final_bench.json")).read().strip().splitlines()[-1]
d = json.loads(line)
open(os.path.join(ROOT, "profiles", f"{tag}_bench_final_n1e6.json"), "w").write(line + "\n")
shas = {f: json.load(open(os.path.join(ROOT, "profiles", f))).get("csrc_sha") for f in
        (f"{tag}_workload_counters.json", f"{tag}_sq_counters_n1e6.json", f"{tag}_pmc_fetch_write_n1e6.json", f"{tag}_pmc_fetch_write_n1e7.json",
         f"{tag}_rays_sq_counters.json")}
print("kernel sources:", csrc_sha(), "profiles:", shas)
print("bench line: %.4f ms/step = %.0f %s; 1e7: %.4f ms/step" % (d["ms_per_step"], d["value"], d["unit"], d["north_star_1e7"]["ms_per_step"]))
