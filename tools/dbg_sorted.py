#!/usr/bin/env python3
"""Run ON THE GPU BOX: build of an input that is already in Morton order (per-kernel times + correctness)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from bench import collect_profile
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
b = ibvh.BVH(v)
sv = b.leaves.volume.contiguous()
b2 = ibvh.BVH(sv)
assert b2.leaves.volume.equal(b.leaves.volume) and bool((b2.leaves.index == torch.arange(1, n + 1, device="cuda", dtype=b2.leaves.index.dtype)).all())
for src, name in ((v, "random order"), (sv, "morton order")):
    c = None
    for _ in range(3):
        c = ibvh.BVH(src, cache=c)
    lib.call("ibvh_profile_enable", 1)
    for _ in range(5):
        c = ibvh.BVH(src, cache=c)
    torch.cuda.synchronize()
    p = collect_profile(lib)
    lib.call("ibvh_profile_enable", 0)
    print(name, {k: round(v_[0] / 5 * 1e3, 1) for k, v_ in p.items()}, "two_level flag", int(c._skew[0]))
