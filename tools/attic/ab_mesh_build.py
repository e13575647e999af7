#!/usr/bin/env python3
"""Run ON THE GPU BOX: build time of config 3's mesh (7.2 M-triangle torus surrogate, mesh order = coherent input) and of a
uniform cloud of the same size under forced sort geometries (IBVH tuning knobs msd_bits / msd_cap), with the per-kernel
breakdown.  usage: ab_mesh_build.py "bits:cap,bits:cap,..."  (0 = the plan's own choice)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api, lib
from implicitbvh_amd.synthetic import sphere_radius_law, torus_mesh
from bench import _dominant

variants = (sys.argv[1] if len(sys.argv) > 1 else "0:0,12:0,12:8192,12:4096,11:16384").split(",")
tris = torch.from_numpy(torus_mesh()).cuda()
vols = ibvh.bounding_volumes_from_triangles(tris)
n = int(vols.shape[0])
cloud = ibvh.generate_spheres(n, 42, r0=sphere_radius_law(n))
for name, v in (("mesh", vols), ("uniform", cloud)):
    ref = None
    for var in variants:
        bits, cap = (int(x) for x in var.split(":"))
        lib.set_tuning("msd_bits", bits)
        lib.set_tuning("msd_cap", cap)
        api._shape_memo.clear()
        st = {"b": None}
        def run():
            st["b"] = ibvh.BVH(v, cache=st["b"])
            return st["b"]
        run(); run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        _, _, ks = _dominant(lib, torch, run)
        leaves = st["b"].leaves.buf.clone()
        same = True if ref is None else bool(torch.equal(ref, leaves))
        if ref is None:
            ref = leaves
        print(f"{name} bits={bits} cap={cap}: build {ms:.3f} ms identical {same}  " + " ".join(f"{k.replace('_kernel','')}={x:.3f}" for k, x in sorted(ks.items(), key=lambda kv: -kv[1])[:8]))
lib.set_tuning("msd_bits", 0); lib.set_tuning("msd_cap", 0)
