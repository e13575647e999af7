#!/usr/bin/env python3
"""Run ON THE GPU BOX: config 3 (torus surrogate mesh + 1e6 rays): per-ray hit counts and how unevenly they fall on
the 64 lanes of a wave (a per-lane walk waits for its heaviest ray)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import implicitbvh_amd as ibvh
from implicitbvh_amd.synthetic import torus_mesh
tris = torch.from_numpy(torus_mesh()).cuda()
vols = ibvh.bounding_volumes_from_triangles(tris)
bvh = ibvh.BVH(vols)
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(43)
hv = vols.cpu().numpy()
lo, hi = hv[:, :3].min(0), hv[:, :3].max(0)
p = torch.from_numpy((lo + (hi - lo) * rng.random((nr, 3))).astype(np.float32)).cuda().t()
d = torch.from_numpy(rng.random((nr, 3)).astype(np.float32)).cuda().t()
t = ibvh.traverse_rays(bvh, p, d)
inc = t.cache2[:nr].cpu().numpy().astype(np.int64)
hits = np.diff(np.concatenate([[0], inc]))
w = hits[: nr // 64 * 64].reshape(-1, 64)
print("rays", nr, "hits", hits.sum(), "mean", hits.mean(), "max", hits.max(), "zero-hit rays %.2f" % (hits == 0).mean())
print("per wave: mean of max %.1f, mean of mean %.2f -> lane utilisation by hits %.2f" % (w.max(1).mean(), w.mean(1).mean(), w.mean(1).mean() / w.max(1).mean()))
st = {"t": t}
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    st["t"] = ibvh.traverse_rays(bvh, p, d, cache=st["t"])
torch.cuda.synchronize()
print("traverse_rays %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
