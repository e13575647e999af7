#!/usr/bin/env python3
"""Run ON THE GPU BOX: per-step source-queue sizes and checks of the breadth-first self traversal of config 2 (1e6
random spheres) beside the traversal's time.  usage: python tools/dbg_bfs_levels.py [n]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
b = ibvh.BVH(v)
w = ibvh.traverse(b, ibvh.BFSTraversal())
w = ibvh.traverse(b, ibvh.BFSTraversal(), cache=w)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    w = ibvh.traverse(b, ibvh.BFSTraversal(), cache=w); w.num_contacts
torch.cuda.synchronize()
print("traversal %.3f ms, contacts %d, checks %d, start level %d of %d" % ((time.perf_counter() - t0) / 10 * 1e3, w.num_contacts, w.num_checks, w.start_level1, b.tree.levels))
c, levels = api._last_bfs_counters
c = c.view(torch.int64).cpu().tolist()
chk = levels + 8
slots = len(c) // chk - 1  # (GEN_SLOTS of ibvh_bfs.hip)
for s in range(levels + 4):
    checks = sum(c[chk * (1 + k) + s] for k in range(slots))
    if c[1 + s] or checks:
        print("step %2d  source pairs %10d  checks %10d  produced %10d" % (s, c[1 + s], checks, c[2 + s]))
