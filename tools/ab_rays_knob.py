#!/usr/bin/env python3
"""GPU box: config 3's 1e6 rays (binned path) under settings of one tuning knob: time per call, kernels, every list == the first.
usage: python tools/ab_rays_knob.py KNOB v0,v1,..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api, lib
from implicitbvh_amd.synthetic import random_rays, torus_mesh
from bench import _dominant
knob = sys.argv[1]
vals = [int(x) for x in sys.argv[2].split(",")]
tris = torch.from_numpy(torus_mesh()).cuda()
vols = ibvh.bounding_volumes_from_triangles(tris)
bvh = ibvh.BVH(vols)
hv = vols[:, :3]
lo, hi = hv.min(0).values.cpu().numpy(), hv.max(0).values.cpu().numpy()
ph, dh = random_rays(1_000_000, lo, hi, seed=43)
p, d = torch.from_numpy(ph).cuda().t(), torch.from_numpy(dh).cuda().t()
first = None
for v in vals:
    lib.set_tuning(knob, v)
    api._shape_memo.clear()
    st = {"t": None}
    def run():
        st["t"] = ibvh.traverse_rays(bvh, p, d, cache=st["t"])
        return st["t"]
    run(); run(); run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        run().num_contacts
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    _, _, ks = _dominant(lib, torch, run)
    c = st["t"].contacts
    same = True if first is None else bool(torch.equal(first, c))
    if first is None:
        first = c.clone()
    print(f"{knob}={v}: {ms:.3f} ms hits {st['t'].num_contacts} identical_to_first {same} " + " ".join(f"{k.replace('_kernel','')}={x:.3f}" for k, x in sorted(ks.items(), key=lambda kv: -kv[1])[:6]), flush=True)
