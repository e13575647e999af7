#!/usr/bin/env python3
"""Run ON THE GPU BOX: config 4's pair traversal (two 5e6-leaf clouds, 10 % overlap), leaf-vs-tree, timed alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd.synthetic import sphere_radius_law
n = 5_000_000
a = ibvh.generate_spheres(n, 42, r0=sphere_radius_law(2 * n))
b = ibvh.generate_spheres(n, 43, r0=sphere_radius_law(2 * n))
b[:, 0] += 0.9
ba, bb = ibvh.BVH(a), ibvh.BVH(b)
t = None
for _ in range(5):
    t = ibvh.traverse(ba, bb, cache=t); t.num_contacts
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        t = ibvh.traverse(ba, bb, cache=t); t.num_contacts
    torch.cuda.synchronize()
    print("pair lvt %.4f ms contacts %d" % ((time.perf_counter() - t0) / 20 * 1e3, t.num_contacts))
