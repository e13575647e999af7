#!/usr/bin/env python3
"""Run ON THE GPU BOX: config 3's rays (7.2 M-triangle torus surrogate, 1e6 random rays) through the breadth-first
traversal, per-step queue sizes included, beside the leaf-vs-tree traversal."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api
from implicitbvh_amd.synthetic import torus_mesh
tris = torch.from_numpy(torus_mesh()).cuda()
vols = ibvh.bounding_volumes_from_triangles(tris)
bvh = ibvh.BVH(vols)
nr = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(43)
hv = vols.cpu().numpy()
lo, hi = hv[:, :3].min(0), hv[:, :3].max(0)
p = torch.from_numpy((lo + (hi - lo) * rng.random((nr, 3))).astype(np.float32)).cuda().t()
d = torch.from_numpy(rng.random((nr, 3)).astype(np.float32)).cuda().t()
for name, alg in (("lvt", ibvh.LVTTraversal()), ("bfs", ibvh.BFSTraversal())):
    t = None
    for _ in range(3):
        t = ibvh.traverse_rays(bvh, p, d, alg, cache=t); t.num_contacts
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        t = ibvh.traverse_rays(bvh, p, d, alg, cache=t); t.num_contacts
    torch.cuda.synchronize()
    print(name, "%.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), "hits", t.num_contacts, "start level", t.start_level1, "of", bvh.tree.levels)
c, levels = api._last_bfs_counters
c = c.view(torch.int64).cpu().tolist()
chk = levels + 8
slots = len(c) // chk - 1
for s in range(levels + 4):
    checks = sum(c[chk * (1 + k) + s] for k in range(slots))
    if c[1 + s] or checks:
        print("step %2d  source pairs %10d  checks %10d  produced %10d" % (s, c[1 + s], checks, c[2 + s]))
