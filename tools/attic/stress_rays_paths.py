#!/usr/bin/env python3
"""Run ON THE GPU BOX: differential stress of the two ray paths.  The per-lane walker (knob rays_binned = 0) is pinned to the
oracle by the parity tests; here the shipped rule (1) and the forced binned path (2, random subtree depths) must return the
SAME contact list, element for element, on random trees (1e3 .. 3e6 leaves, spheres / boxes, Float32 / Float64, clustered and
uniform), random ray batches (1 .. 2e5 rays, with irregular rays mixed in), random start levels and the ray narrow.
usage: stress_rays_paths.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api, lib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    n = int(10 ** rng.uniform(3, 6.5))
    nr = int(10 ** rng.uniform(0, 5.3))
    f64 = bool(rng.integers(0, 4) == 0)
    boxes = bool(rng.integers(0, 2))
    npdt, tdt = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
    scale = float(rng.uniform(3, 80))
    ctr = rng.random((n, 3)) * scale
    if rng.integers(0, 3) == 0:  # clustered
        ctr = (rng.random((8, 3)) * scale)[rng.integers(0, 8, n)] + rng.normal(0, scale / 60, (n, 3))
    size = rng.uniform(0.05, 1.2)
    if boxes:
        h = size * (0.1 + 0.9 * rng.random((n, 3)))
        vols = np.concatenate([ctr - h, ctr + h], axis=1).astype(npdt)
        leaf = ibvh.BBox(tdt)
    else:
        vols = np.concatenate([ctr, size * (0.1 + 0.9 * rng.random((n, 1)))], axis=1).astype(npdt)
        leaf = None
    node = ibvh.BBox(tdt) if boxes or rng.integers(0, 3) else ibvh.BSphere(tdt)
    bvh = ibvh.BVH(torch.from_numpy(vols).cuda(), node)
    p = (rng.random((nr, 3)) * (scale + 4) - 2).astype(npdt)
    d = (rng.random((nr, 3)) - 0.5).astype(npdt)
    if nr > 10:
        d[::7, rng.integers(0, 3)] = 0
        d[3::31] = 0
        p[5::43, 0] = np.nan
        d[9::53, 1] = np.inf
    P, D = torch.from_numpy(p).cuda().t(), torch.from_numpy(d).cuda().t()
    sl = int(rng.integers(1, bvh.tree.levels + 1)) if rng.integers(0, 3) == 0 else 1
    narrow = ibvh.NARROW_RAY_ORIGIN_OUTSIDE if rng.integers(0, 4) == 0 else None
    out = {}
    for mode, depth in ((0, 0), (1, 0), (2, int(rng.integers(1, 12)))):
        lib.set_tuning("rays_binned", mode)
        lib.set_tuning("rays_subtree_depth", depth)
        api._shape_memo.clear()
        t = ibvh.traverse_rays(bvh, P, D, start_level=sl, narrow=narrow)
        t2 = ibvh.traverse_rays(bvh, P, D, start_level=sl, narrow=narrow, cache=t)
        assert torch.equal(t.contacts, t2.contacts)
        out[mode] = t.contacts.clone()
    ok = torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])
    bad += 0 if ok else 1
    print(f"case {c}: leaves {n} levels {bvh.tree.levels} rays {nr} f64 {f64} boxes {boxes} node {type(node).__name__} start {sl} narrow {narrow is not None} hits {out[0].shape[0]} {'ok' if ok else 'MISMATCH'}")
lib.set_tuning("rays_binned", 1); lib.set_tuning("rays_subtree_depth", 0)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
