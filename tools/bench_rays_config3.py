"""config 3's ray traversal alone (7.2 M-triangle torus, 1e6 random rays): ms per call of a `cache=` chain + the library's per-kernel
events.  usage: [IBVH_TUNING=rays_tail=0] python tools/bench_rays_config3.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import lib  # noqa: E402
from implicitbvh_amd.synthetic import random_rays, torus_mesh  # noqa: E402

vols = ibvh.bounding_volumes_from_triangles(torch.from_numpy(torus_mesh()).cuda())
b = ibvh.BVH(vols)
hv = vols[:, :3]
lo, hi = hv.min(0).values.cpu().numpy(), hv.max(0).values.cpu().numpy()
ph, dh = random_rays(1_000_000, lo, hi, seed=43)
p, d = torch.from_numpy(ph).cuda().t(), torch.from_numpy(dh).cuda().t()
st = {"r": None}


def rays():
    st["r"] = ibvh.traverse_rays(b, p, d, cache=st["r"])
    return st["r"]
for _ in range(4):
    rays().num_contacts
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    rays().num_contacts
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
_, _, ks = bench._dominant(lib, torch, rays)
print(f"traverse_rays {ms:.4f} ms, {st['r'].num_contacts} hits  " + " ".join(f"{k.replace('_kernel', '')}={v:.3f}" for k, v in ks.items() if v > 0.02))
