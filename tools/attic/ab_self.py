#!/usr/bin/env python3
"""Run ON THE GPU BOX: the dense leaf-query workloads — config 3's self-traversal (7.2 M-triangle surrogate, 82 M contacts) and
config 4's pair traversal (two 5e6-leaf clouds) — per-call time and per-kernel breakdown of a `cache=` chain."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from implicitbvh_amd.synthetic import sphere_radius_law, torus_mesh
from bench import _dominant


def report(name, fn):
    fn(); fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        r = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    _, _, ks = _dominant(lib, torch, fn)
    print(f"{name}: {ms:.4f} ms contacts {r.num_contacts}  " + " ".join(f"{k}={v:.4f}" for k, v in ks.items()))


vols = ibvh.bounding_volumes_from_triangles(torch.from_numpy(torus_mesh()).cuda())
b = ibvh.BVH(vols)
st = {"t": None}
def self3():
    st["t"] = ibvh.traverse(b, cache=st["t"]); return st["t"]
report("config3 self", self3)
del b, vols; st["t"] = None
n = 5_000_000
r0 = sphere_radius_law(n)
b1 = ibvh.BVH(ibvh.generate_spheres(n, 44, r0=r0)); b2 = ibvh.BVH(ibvh.generate_spheres(n, 45, origin=(0.9, 0.0, 0.0), r0=r0))
def pair4():
    st["t"] = ibvh.traverse(b1, b2, cache=st["t"]); return st["t"]
report("config4 pair", pair4)
