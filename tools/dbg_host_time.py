#!/usr/bin/env python3
"""Run ON THE GPU BOX: is the bench step bound by the GPU or by the host that enqueues it?  Time to ISSUE 200 steps vs
time until the GPU has finished them."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicitbvh_amd as ibvh
for n in (250_000, 500_000, 1_000_000, 2_000_000):
    v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
    st = (None, None)
    for _ in range(30):
        b = ibvh.BVH(v, cache=st[0]); st = (b, ibvh.traverse(b, cache=st[1]))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        b = ibvh.BVH(v, cache=st[0]); st = (b, ibvh.traverse(b, cache=st[1]))
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(n, "issue %.1f us/step, done %.1f us/step" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
