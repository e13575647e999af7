#!/usr/bin/env python3
"""Run ON THE GPU BOX: LVT self-traversal of 1e6 / 1e7 random spheres, config 3's mesh and config 4's pair: per-call time of a
`cache=` chain and the per-kernel events (count pass, scans, write pass)."""
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
    for _ in range(10):
        r = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    _, _, ks = _dominant(lib, torch, fn)
    print(f"{name}: {ms:.4f} ms contacts {r.num_contacts}  " + " ".join(f"{k.replace('_kernel','')}={v:.4f}" for k, v in ks.items()), flush=True)
    return r


st = {"t": None}
for n in (1_000_000, 10_000_000):
    b = ibvh.BVH(ibvh.generate_spheres(n, 42, r0=sphere_radius_law(n)))
    def self_():
        st["t"] = ibvh.traverse(b, cache=st["t"]); return st["t"]
    report(f"self {n}", self_)
    st["t"] = None
    del b
b = ibvh.BVH(ibvh.bounding_volumes_from_triangles(torch.from_numpy(torus_mesh()).cuda()))
def self3():
    st["t"] = ibvh.traverse(b, cache=st["t"]); return st["t"]
report("config3 self", self3)
del b; st["t"] = None
n = 5_000_000
r0 = sphere_radius_law(n)
b1 = ibvh.BVH(ibvh.generate_spheres(n, 44, r0=r0)); b2 = ibvh.BVH(ibvh.generate_spheres(n, 45, origin=(0.9, 0.0, 0.0), r0=r0))
def pair4():
    st["t"] = ibvh.traverse(b1, b2, cache=st["t"]); return st["t"]
report("config4 pair", pair4)
