import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api, lib
from implicitbvh_amd.synthetic import sphere_radius_law
from bench import _dominant
n = 5_000_000
r0 = sphere_radius_law(n)
b1 = ibvh.BVH(ibvh.generate_spheres(n, 44, r0=r0)); b2 = ibvh.BVH(ibvh.generate_spheres(n, 45, origin=(0.9, 0.0, 0.0), r0=r0))
m = 1_000_000
bs = ibvh.BVH(ibvh.generate_spheres(m, 42, r0=sphere_radius_law(m)))
for K in (8, 16, 32):
    api.LVT_CACHE_SLOTS = K
    api._shape_memo.clear()
    st = {"t": None, "s": None}
    def pair4():
        st["t"] = ibvh.traverse(b1, b2, cache=st["t"]); return st["t"]
    def self2():
        st["s"] = ibvh.traverse(bs, cache=st["s"]); return st["s"]
    for name, fn in (("pair", pair4), ("self1e6", self2)):
        fn(); fn(); fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): r = fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        _, _, ks = _dominant(lib, torch, fn)
        print(K, name, round(ms, 4), ks)
