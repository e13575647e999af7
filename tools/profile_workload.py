#!/usr/bin/env python3
"""Run ON THE GPU BOX under rocprofv3: exactly ONE workload of BASELINE.json's configs, so that every launch of its kernels in
the profile belongs to it (the old per-script profile averaged a kernel over rays of several sizes and BFS levels
of every config).  Set-up (generators, builds) launches other kernels only.
usage: python3 tools/profile_workload.py NAME [reps]
NAME: config2_lvt | config2_bfs | config3_self | config3_rays | config3_rays_bfs | config4_pair_lvt | config4_pair_bfs |
      timestep_1e6 | timestep_1e7 | config2_f64nodes"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd.synthetic import random_rays, sphere_radius_law, torus_mesh

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
st = {"t": None}
info = {"workload": name, "reps": reps}
if name.startswith("config2"):
    n = 1_000_000
    v = ibvh.generate_spheres(n, 42, r0=sphere_radius_law(n))
    if name == "config2_f64nodes":
        b = ibvh.BVH(v.double(), ibvh.BBox(torch.float64))
    else:
        b = ibvh.BVH(v)
    alg = ibvh.BFSTraversal() if name == "config2_bfs" else ibvh.LVTTraversal()

    def fn():
        st["t"] = ibvh.traverse(b, alg, cache=st["t"])
        return st["t"]
    info["leaves"] = n
elif name.startswith("config3"):
    tris = torch.from_numpy(torus_mesh()).cuda()
    vols = ibvh.bounding_volumes_from_triangles(tris)
    del tris
    b = ibvh.BVH(vols)
    info["leaves"] = int(vols.shape[0])
    if name == "config3_self":
        def fn():
            st["t"] = ibvh.traverse(b, cache=st["t"])
            return st["t"]
    else:
        hv = vols[:, :3]
        lo, hi = hv.min(0).values.cpu().numpy(), hv.max(0).values.cpu().numpy()
        nr = 1_000_000
        p_host, d_host = random_rays(nr, lo, hi, seed=43)
        p, d = torch.from_numpy(p_host).cuda().t(), torch.from_numpy(d_host).cuda().t()
        alg = ibvh.BFSTraversal() if name == "config3_rays_bfs" else ibvh.LVTTraversal()
        info["rays"] = nr

        def fn():
            st["t"] = ibvh.traverse_rays(b, p, d, alg, cache=st["t"])
            return st["t"]
elif name.startswith("config4"):
    n = 5_000_000
    r0 = sphere_radius_law(n)
    b1 = ibvh.BVH(ibvh.generate_spheres(n, 44, r0=r0))
    b2 = ibvh.BVH(ibvh.generate_spheres(n, 45, origin=(0.9, 0.0, 0.0), r0=r0))
    alg = ibvh.BFSTraversal() if name == "config4_pair_bfs" else ibvh.LVTTraversal()
    info["leaves_each"] = n

    def fn():
        st["t"] = ibvh.traverse(b1, b2, alg, cache=st["t"])
        return st["t"]
elif name.startswith("timestep"):
    n = 1_000_000 if name.endswith("1e6") else 10_000_000
    bv = ibvh.BoundingVolumes.wrap(ibvh.generate_spheres(n, 47, r0=sphere_radius_law(n)), torch.arange(n, 0, -1, dtype=torch.int32, device="cuda"))
    g = torch.Generator(device="cuda").manual_seed(11)
    st["b"] = ibvh.BVH(bv)
    info["leaves"] = n

    def fn():
        st["b"].leaves.volume[:, :3] += (torch.rand((n, 3), generator=g, device="cuda") * 2 - 1) / 1024.0
        st["b"] = ibvh.BVH(st["b"].leaves, cache=st["b"])
        st["t"] = ibvh.traverse(st["b"], cache=st["t"])
        return st["t"]
else:
    raise SystemExit("unknown workload " + name)
for _ in range(2):
    fn().num_contacts
torch.cuda.synchronize()
for _ in range(reps):
    c = fn().num_contacts
torch.cuda.synchronize()
info["contacts"] = int(c)
if hasattr(st["t"], "num_checks"):
    info["num_checks"] = int(st["t"].num_checks or 0)
print(json.dumps(info))
