#!/usr/bin/env python3
"""Run ON THE GPU BOX: wall time of each of the first steps of the bench loop (build + traverse chained through cache=),
each step synchronised — how long until the steady state?"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicitbvh_amd as ibvh
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
state = (None, None)
times = []
for it in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = ibvh.BVH(v, cache=state[0])
    t = ibvh.traverse(b, cache=state[1])
    state = (b, t)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
print(n, " ".join("%.2f" % x for x in times))
# unsynchronised run of 8 after that
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(8):
    b = ibvh.BVH(v, cache=state[0]); t = ibvh.traverse(b, cache=state[1]); state = (b, t)
torch.cuda.synchronize()
print("8 chained steps: %.3f ms each" % ((time.perf_counter() - t0) / 8 * 1e3))
