#!/usr/bin/env python3
"""Run ON THE GPU BOX: what the idle launches of the extra partition levels cost a uniform build (api.COLD_SORT_LEVELS)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api
for n in (1_000_000, 10_000_000):
    v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
    b = ibvh.BVH(v)
    for levels in (0, 1, 2, 4, 0, 2, 4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            b._skew[0] = levels - 1 if levels else 0   # the cached build launches (value + 1) levels
            b = ibvh.BVH(v, cache=b)
        torch.cuda.synchronize()
        print(n, "levels", levels, "build %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
