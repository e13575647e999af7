#!/usr/bin/env python3
"""Run ON THE GPU BOX: build at several sizes / knob settings, each in its own process (a GPU fault kills only it)."""
import math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = f'''
import sys, math
sys.path.insert(0, {ROOT!r})
import torch, implicitbvh_amd as ibvh
n=int(sys.argv[1])
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
b = ibvh.BVH(v)
torch.cuda.synchronize()
m = b.leaves.morton
assert bool((m[1:] >= m[:-1]).all()), "not sorted"
idx = b.leaves.index.cpu()
assert idx.sort().values.equal(torch.arange(1, n+1, dtype=idx.dtype)), "not a permutation"
print("ok", n)
'''
for spec in sys.argv[1:]:
    n, *kv = spec.split(",")
    env = dict(os.environ)
    for x in kv:
        k, v = x.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", code, n], env=env, capture_output=True, text=True, timeout=300)
    print(spec, "->", (r.stdout.strip() or "FAILED rc=%d %s" % (r.returncode, r.stderr.strip()[-300:])), flush=True)
