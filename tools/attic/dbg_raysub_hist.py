#!/usr/bin/env python3
"""GPU box, IBVH_LIB=variants/libibvh_rsh.so (tools/build_variant.sh rsh -DIBVH_RAYSUB_HIST): wave-steps of rays_subtree_kernel's
counting pass on config 3 by number of busy lanes, and how many of them come after a workgroup's chunk ran dry."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from implicitbvh_amd.synthetic import random_rays, torus_mesh
tris = torch.from_numpy(torus_mesh()).cuda()
vols = ibvh.bounding_volumes_from_triangles(tris)
bvh = ibvh.BVH(vols)
hv = vols[:, :3]
lo, hi = hv.min(0).values.cpu().numpy(), hv.max(0).values.cpu().numpy()
ph, dh = random_rays(1_000_000, lo, hi, seed=43)
p, d = torch.from_numpy(ph).cuda().t(), torch.from_numpy(dh).cuda().t()
t = ibvh.traverse_rays(bvh, p, d)
torch.cuda.synchronize()
L = lib.load()
L.ibvh_debug_raysub_hist.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(16, dtype=np.uint64)
L.ibvh_debug_raysub_hist(buf.ctypes.data, 1)
t = ibvh.traverse_rays(bvh, p, d)
torch.cuda.synchronize()
L.ibvh_debug_raysub_hist(buf.ctypes.data, 0)
tot = float(buf[:8].sum())
print("hits", t.num_contacts, "wave-steps", int(tot))
for k in range(8):
    print(f"  busy lanes {8*k:2d}..{8*k+7 if k < 7 else 64:2d}: {100*float(buf[k])/tot:5.1f} %")
print(f"  after the chunk ran dry: {100*float(buf[8]+buf[9])/tot:5.1f} % of the steps ({100*float(buf[8])/tot:.1f} % with < 8 lanes busy)")
