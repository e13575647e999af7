#!/usr/bin/env python3
"""Run ON THE GPU BOX: BFS without cache= (queues allocated, grown, resumed) against the warm traversal, for several
growth factors of the queues."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api, lib
n = 1_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
b = ibvh.BVH(v)
calls = {"n": 0}
orig = lib.load().ibvh_traverse_bfs
for growth in (2, 4, 8):
    api.BFS_GROWTH = growth
    for _ in range(2):
        t = ibvh.traverse(b, ibvh.BFSTraversal()); t.num_contacts
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        t = ibvh.traverse(b, ibvh.BFSTraversal()); t.num_contacts
    torch.cuda.synchronize()
    print("growth", growth, "no cache %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), "queue capacity", t.cache1.shape[0], "contacts", t.num_contacts)
w = ibvh.traverse(b, ibvh.BFSTraversal())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    w = ibvh.traverse(b, ibvh.BFSTraversal(), cache=w); w.num_contacts
torch.cuda.synchronize()
print("with cache %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
