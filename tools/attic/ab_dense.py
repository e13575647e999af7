#!/usr/bin/env python3
"""Run ON THE GPU BOX: dual descent (lvt_dual=1) against round 3's lvt_queue_kernel (0) on the DENSE workloads: config 3's
mesh self-traversal (7.2 M-triangle torus surrogate, ~11 contacts per leaf) and config 4's pair traversal (two 5e6-leaf
clouds, 10 % overlap); contact lists compared byte for byte.  usage: [IBVH_LIB=...] python tools/ab_dense.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from implicitbvh_amd.synthetic import sphere_radius_law


def timed(fn, reps=10):
    for _ in range(3):
        t = fn()
        t.num_contacts
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        t = fn()
        t.num_contacts
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, t


from implicitbvh_amd.synthetic import torus_mesh
mesh = torch.from_numpy(torus_mesh()).cuda()
cases = []
if mesh is not None:
    vols = ibvh.bounding_volumes_from_triangles(mesh)
    bm = ibvh.BVH(vols)
    st = {"t": None}

    def self3():
        st["t"] = ibvh.traverse(bm, cache=st["t"])
        return st["t"]
    cases.append(("config3 self", self3))
n = 5_000_000
a = ibvh.generate_spheres(n, 44, r0=sphere_radius_law(n))
b = ibvh.generate_spheres(n, 45, r0=sphere_radius_law(n))
b[:, 0] += 0.9
ba, bb = ibvh.BVH(a), ibvh.BVH(b)
sp = {"t": None}


def pair4():
    sp["t"] = ibvh.traverse(ba, bb, cache=sp["t"])
    return sp["t"]


cases.append(("config4 pair", pair4))
for name, fn in cases:
    ref = None
    for dual in (1, 0):
        lib.set_tuning("lvt_dual", dual)
        ms, t = timed(fn)
        c = t.contacts.clone()
        if ref is None:
            ref = c
        print(f"{name} dual={dual} {ms:.4f} ms contacts {t.num_contacts} identical {ref.shape == c.shape and bool(torch.equal(ref, c))}", flush=True)
