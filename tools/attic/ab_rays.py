#!/usr/bin/env python3
"""Run ON THE GPU BOX: config 3 (7.2 M-triangle torus surrogate, 1e6 random rays): traverse_rays with the binned path
(knob rays_binned = 1, subtree depths 9 .. 11) against the binary walker (0) and the breadth-first algorithm — per-call
time, per-kernel breakdown (library profiler), and that every variant returns the same list."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api, lib
from implicitbvh_amd.synthetic import random_rays, torus_mesh

nr = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0:0", "1:10", "1:9", "1:11", "1:8"]
uv = os.environ.get("AB_MESH_UV", "")  # e.g. "353,354": the 249,924-triangle surrogate of the reference's published case
tris = torch.from_numpy(torus_mesh(*[int(x) for x in uv.split(",")]) if uv else torus_mesh()).cuda()
vols = ibvh.bounding_volumes_from_triangles(tris)
f64 = os.environ.get("AB_F64", "") == "1"  # the same mesh and rays in double precision (BBox{Float64} nodes)
bvh = ibvh.BVH(vols.double(), ibvh.BBox(torch.float64)) if f64 else ibvh.BVH(vols)
print("leaves", int(vols.shape[0]), "levels", bvh.tree.levels, "rays", nr)
hv = vols[:, :3]
lo, hi = hv.min(0).values.cpu().numpy(), hv.max(0).values.cpu().numpy()
ph, dh = random_rays(nr, lo, hi, seed=43)
p, d = torch.from_numpy(ph).cuda().t(), torch.from_numpy(dh).cuda().t()
if f64:
    p, d = p.double(), d.double()
first = None
for v in variants:
    mode, depth = (int(x) for x in v.split(":")[:2])
    per_ray = int(v.split(":")[2]) if v.count(":") >= 2 else 0
    try:
        lib.set_tuning("rays_fast_slab", int(v.split(":")[3]) if v.count(":") >= 3 else 1)
    except Exception:  # noqa: BLE001  (a development build without the knob)
        pass
    lib.set_tuning("rays_binned", mode)
    lib.set_tuning("rays_subtree_depth", depth)
    lib.set_tuning("rays_items_per_ray", per_ray)
    api._shape_memo.clear()
    st = {"t": None}
    def run():
        st["t"] = ibvh.traverse_rays(bvh, p, d, cache=st["t"])
    run(); run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    from bench import _dominant
    _, _, ks = _dominant(lib, torch, run)
    c = st["t"].contacts
    same = True if first is None else bool(torch.equal(first, c))
    if first is None:
        first = c.clone()
    print(f"binned={mode} depth={depth} per_ray={per_ray}: {ms:.3f} ms hits {st['t'].num_contacts} identical_to_first {same}")
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]):
        print(f"    {k:40s} {v:.4f} ms")
