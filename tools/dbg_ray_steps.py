#!/usr/bin/env python3
"""Run ON THE GPU BOX with IBVH_LIB=tools/libibvh_stamps.so built with -DIBVH_RAY_STEPS: distribution of walk steps per ray."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np, torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib, api
from implicitbvh_amd.synthetic import torus_mesh
tris = torch.from_numpy(torus_mesh()).cuda()
vols = ibvh.bounding_volumes_from_triangles(tris)
bvh = ibvh.BVH(vols)
nr = 1_000_000
rng = np.random.default_rng(43)
hv = vols.cpu().numpy()
lo, hi = hv[:, :3].min(0), hv[:, :3].max(0)
p = torch.from_numpy((lo + (hi - lo) * rng.random((nr, 3))).astype(np.float32)).cuda().contiguous()
d = torch.from_numpy(rng.random((nr, 3)).astype(np.float32)).cuda().contiguous()
counts = torch.zeros(nr, dtype=torch.int32, device="cuda")
need = C.c_size_t()
lib.call("ibvh_lvt_scratch_bytes", C.byref(bvh.types), nr, 0, C.byref(need))
scratch = torch.zeros(need.value, dtype=torch.uint8, device="cuda")
s = bvh.struct()
# count pass only; the scan turns counts into an inclusive prefix: take differences
total = C.c_int64()
lib.call("ibvh_traverse_rays_lvt_count", C.byref(s), p.data_ptr(), d.data_ptr(), nr, 1, counts.data_ptr(), C.byref(total), scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream)
inc = counts.cpu().numpy().astype(np.int64)
steps = np.diff(np.concatenate([[0], inc]))
print("rays", nr, "steps total %.3e mean %.1f median %.0f p99 %.0f max %d" % (steps.sum(), steps.mean(), np.median(steps), np.percentile(steps, 99), steps.max()))
w = steps[: nr // 256 * 256].reshape(-1, 256)
print("per block of 256 rays: mean of sum %.0f, max of sum %.0f" % (w.sum(1).mean(), w.sum(1).max()))
