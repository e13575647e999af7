#!/usr/bin/env python3
"""Run ON THE GPU BOX: cost of a build without cache= (allocations, pinned hint word, extra partition levels)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
for levels in (2, 0, 1, 2, 4):
    api.COLD_SORT_LEVELS = levels
    for _ in range(3): b = ibvh.BVH(v)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): b = ibvh.BVH(v)
    torch.cuda.synchronize()
    print(n, "cold build, levels", levels, "%.1f us" % ((time.perf_counter() - t0) / 30 * 1e6))
b = ibvh.BVH(v)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): b = ibvh.BVH(v, cache=b)
torch.cuda.synchronize()
print(n, "cached build %.1f us" % ((time.perf_counter() - t0) / 30 * 1e6))
t0 = time.perf_counter()
for _ in range(30): x = torch.full((1,), 2, dtype=torch.int32).pin_memory()
print("pinned word alloc %.1f us" % ((time.perf_counter() - t0) / 30 * 1e6))
