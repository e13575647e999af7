#!/usr/bin/env python3
"""Run ON THE GPU BOX: time the build's sort kernels under different development knobs (ibvh_set_tuning names, e.g.
msd_bits=10,msd_tile=2048; one bench.py child per setting, handed over as IBVH_TUNING, which the Python binding applies
when it loads the library).  usage: python tools/sort_sweep.py N "name=V,name=V" "name=V" ...
Prints one line per setting: Morton+sort phase ms and the per-kernel averages (us)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = sys.argv[1]
for setting in sys.argv[2:] or [""]:
    env = dict(os.environ, IBVH_TUNING=setting)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", n, "--no-cpu-baseline", "--extra-n", "0",
                          "--steps", "5", "--warmup", "2"], env=env, capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print(setting, "FAILED", out.stderr[-400:])
        continue
    d = json.loads(lines[-1])
    ks = d["kernels"]
    phase = d["roofline"].get("morton_sort_phase", {})
    build = {k: round(v["avg_ms"] * 1e3 * v["launches_per_step"], 1) for k, v in ks.items() if not k.startswith(("lvt_", "scan_reduce", "scan_apply"))}
    print(f"{setting or 'default':40s} phase {phase.get('ms')} ms frac {phase.get('frac')}  step {d['ms_per_step']} ms  {build}", flush=True)
