#!/usr/bin/env python3
"""Run ON THE GPU BOX: build time with the plain Morton grid (+ extra partition levels) against equalised cells
(ibvh_build_desc.sort_equalize; api.EQUALIZE) on config 3's mesh surrogate, a uniform cloud of that size, 1e6 / 1e7 leaves
uniform and in 8 tight Gaussian clusters, 1,000 distinct centres, one cluster + outlier; `cache=` chains (the hint decides the
route and the levels), per-kernel breakdown, results compared byte for byte."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api, lib
from implicitbvh_amd.synthetic import sphere_radius_law, torus_mesh
from bench import _dominant

which = sys.argv[1].split(",") if len(sys.argv) > 1 else None
api.EQ_SPARE_OCCUPANCY = int(os.environ.get("EQ_SPARE", api.EQ_SPARE_OCCUPANCY))  # (experiments: 129 = an equalised build never launches a spare level)
def clustered(n, seed=7):
    g = torch.Generator(device="cuda").manual_seed(seed)
    c = torch.rand((8, 3), generator=g, device="cuda")
    v = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    v[:, :3] = c[torch.randint(0, 8, (n,), generator=g, device="cuda")] + 0.004 * torch.randn((n, 3), generator=g, device="cuda")
    v[:, 3] = 1e-4
    return v
def centres(n, k=1000):
    g = torch.Generator(device="cuda").manual_seed(9)
    c = torch.rand((k, 3), generator=g, device="cuda")
    v = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    v[:, :3] = c[torch.randint(0, k, (n,), generator=g, device="cuda")]
    v[:, 3] = 1e-4
    return v
def outlier(n):
    g = torch.Generator(device="cuda").manual_seed(11)
    v = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    v[:, :3] = 1e-4 * torch.randn((n, 3), generator=g, device="cuda")
    v[n // 2, :3] = 1000.0
    v[:, 3] = 1e-4
    return v
def mesh():
    return ibvh.bounding_volumes_from_triangles(torch.from_numpy(torus_mesh()).cuda())
cases = {"mesh": mesh, "uniform_7.2M": lambda: ibvh.generate_spheres(7_200_000, 42, r0=sphere_radius_law(7_200_000)),
         "uniform_1e6": lambda: ibvh.generate_spheres(1_000_000, 42), "clusters_1e6": lambda: clustered(1_000_000),
         "uniform_1e7": lambda: ibvh.generate_spheres(10_000_000, 42), "clusters_1e7": lambda: clustered(10_000_000),
         "centres_1e6": lambda: centres(1_000_000), "outlier_1e6": lambda: outlier(1_000_000)}
for name, make in cases.items():
    if which and name not in which:
        continue
    v = make()
    ref = None
    for eq in (False, True):
        api.EQUALIZE = eq
        lib.set_tuning("msd_equalize", 0)
        st = {"b": None}
        def run():
            st["b"] = ibvh.BVH(v, cache=st["b"])
            return st["b"]
        for _ in range(4):
            run(); torch.cuda.synchronize()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        d = st["b"]._fast[1]
        _, _, ks = _dominant(lib, torch, run)
        leaves = st["b"].leaves.buf.clone()
        same = True if ref is None else bool(torch.equal(ref, leaves))
        ref = leaves if ref is None else ref
        print(f"{name} equalize={int(eq)} (asked {d.sort_equalize}, levels {d.sort_levels}, hint {int(api._host_words().words[st['b']._skew.slot]):#x}): "
              f"build {ms:.3f} ms identical {same}  " + " ".join(f"{k.replace('_kernel','')}={x:.3f}" for k, x in sorted(ks.items(), key=lambda kv: -kv[1])[:11]), flush=True)
    del v, ref, st
    torch.cuda.empty_cache()
api.EQUALIZE = True
