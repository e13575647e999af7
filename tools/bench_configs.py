"""Timings of BASELINE.json configs 2 (1e6 spheres: LVT and BFS, cold and cached build), 3 (mesh + 1e6 rays) and 4
(two 5e6 clouds, pair traverse) on one MI355X.
Not the driver's bench line (bench.py): these are the parity-test configurations, timed for DESIGN.md."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from bench import collect_profile


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def kernels(fn):
    lib.call("ibvh_profile_enable", 1)
    fn()
    torch.cuda.synchronize()
    p = collect_profile(lib)
    lib.call("ibvh_profile_enable", 0)
    return {k: round(v[0], 4) for k, v in p.items() if v[0] > 0.01}


out = {}
# ---- config 2: LVT and BFS, build cold and with cache= reuse ---------------------------------
n2 = 1_000_000
v2 = ibvh.generate_spheres(n2, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n2)) ** (1 / 3))
ms_cold, _ = timed(lambda: ibvh.BVH(v2))
s2 = {"bvh": None, "lvt": None, "bfs": None}


def build2():
    s2["bvh"] = ibvh.BVH(v2, cache=s2["bvh"])
    return s2["bvh"]


def lvt2():
    s2["lvt"] = ibvh.traverse(s2["bvh"], cache=s2["lvt"])
    return s2["lvt"]


def bfs2():
    s2["bfs"] = ibvh.traverse(s2["bvh"], ibvh.BFSTraversal(), cache=s2["bfs"])
    return s2["bfs"]


ms_warm, _ = timed(build2, 10)
ms_lvt, t_lvt = timed(lvt2, 10)
ms_bfs2, t_bfs = timed(bfs2, 5)
# the same traversal without cache=: queues allocated at 4x the initial pair count, grown and RESUMED where they overflow
torch.cuda.synchronize()
_t0 = time.perf_counter()
for _ in range(3):
    _c = ibvh.traverse(s2["bvh"], ibvh.BFSTraversal())
    _c.num_contacts
torch.cuda.synchronize()
ms_bfs_cold = (time.perf_counter() - _t0) / 3 * 1e3
del _c
out["config2"] = {"leaves": n2, "build_cold_ms": round(ms_cold, 3), "build_cache_ms": round(ms_warm, 3),
                  "traverse_lvt_ms": round(ms_lvt, 3), "traverse_bfs_ms": round(ms_bfs2, 3), "traverse_bfs_no_cache_ms": round(ms_bfs_cold, 3),
                  "contacts": t_lvt.num_contacts,
                  "bfs_contacts": t_bfs.num_contacts, "bfs_checks": t_bfs.num_checks,
                  "Mcontacts_per_s_lvt": round(t_lvt.num_contacts / ms_lvt / 1e3, 1),
                  "Mcontacts_per_s_bfs": round(t_bfs.num_contacts / ms_bfs2 / 1e3, 1)}
# ---- skew and order: the same 1e6 leaves (a) drawn from 8 tight Gaussian clusters — a handful of Morton
# cells hold everything, the build's second partition level takes them — and (b) uniform but handed over already in
# Morton order (the time-stepping shape: leaves of the previous step fed back)
def checked_build(vols, what):
    b = ibvh.BVH(vols)
    m = b.leaves.morton
    assert bool((m[1:] >= m[:-1]).all()), what + ": Morton codes not ascending"
    idx = b.leaves.index.cpu()
    assert idx.sort().values.equal(torch.arange(1, len(idx) + 1, dtype=idx.dtype)), what + ": not a permutation"
    return b


g = torch.Generator(device="cuda").manual_seed(7)
centres = torch.rand((8, 3), generator=g, device="cuda")
which = torch.randint(0, 8, (n2,), generator=g, device="cuda")
clustered = torch.empty((n2, 4), dtype=torch.float32, device="cuda")
clustered[:, :3] = centres[which] + 0.004 * torch.randn((n2, 3), generator=g, device="cuda")
clustered[:, 3] = 1e-4
checked_build(clustered, "clustered")
sc = {"bvh": None}


def build_clustered():
    sc["bvh"] = ibvh.BVH(clustered, cache=sc["bvh"])
    return sc["bvh"]


ms_clustered, _ = timed(build_clustered, 10)
sorted_vols = checked_build(v2, "uniform").leaves.volume.contiguous()
ss = {"bvh": None}


def build_sorted():
    ss["bvh"] = ibvh.BVH(sorted_vols, cache=ss["bvh"])
    return ss["bvh"]


ms_sorted, _ = timed(build_sorted, 10)
out["config2_skew"] = {"leaves": n2, "build_uniform_ms": round(ms_warm, 3), "build_8_gaussian_clusters_ms": round(ms_clustered, 3),
                       "clustered_over_uniform": round(ms_clustered / ms_warm, 2), "build_morton_sorted_input_ms": round(ms_sorted, 3),
                       "kernels_clustered_ms": kernels(build_clustered)}
del v2, s2, t_lvt, t_bfs, clustered, sorted_vols, sc, ss
torch.cuda.empty_cache()
# ---- config 3 -------------------------------------------------------------------------------
from implicitbvh_amd.synthetic import torus_mesh
# IBVH_MESH=/path/to/xyzrgb_dragon.obj (benchmark/bvh_contact.jl:30-36): the real mesh when it is there,
# the 7.2 M-triangle torus surrogate otherwise
mesh_path = os.environ.get("IBVH_MESH", "")
if mesh_path and os.path.exists(mesh_path):
    tris = ibvh.load_obj_triangles(mesh_path)
    out["config3_mesh"] = mesh_path
else:
    tris = torch.from_numpy(torus_mesh()).cuda()
    out["config3_mesh"] = "torus surrogate (IBVH_MESH not set)"
ms_vol, vols = timed(lambda: ibvh.bounding_volumes_from_triangles(tris))
state = {"bvh": None, "t": None}


def build3():
    state["bvh"] = ibvh.BVH(vols, cache=state["bvh"])
    return state["bvh"]


ms_build, bvh = timed(build3)
nr = 1_000_000
rng = np.random.default_rng(43)
hv = vols.cpu().numpy()
lo, hi = hv[:, :3].min(0), hv[:, :3].max(0)
p = torch.from_numpy((lo + (hi - lo) * rng.random((nr, 3))).astype(np.float32)).cuda().t()
d = torch.from_numpy(rng.random((nr, 3)).astype(np.float32)).cuda().t()


def rays():
    state["t"] = ibvh.traverse_rays(bvh, p, d, cache=state["t"])
    return state["t"]


ms_rays, tr = timed(rays, 3)
out["config3"] = {"triangles": int(tris.shape[0]), "volumes_ms": round(ms_vol, 3), "build_ms": round(ms_build, 3),
                  "build_kernels_ms": kernels(build3),
                  "rays": nr, "traverse_rays_lvt_ms": round(ms_rays, 3), "hits": tr.num_contacts,
                  "Mrays_per_s": round(nr / ms_rays / 1e3, 2), "kernels_ms": kernels(rays)}
state["t"] = None
ms_self, ts = timed(lambda: ibvh.traverse(bvh), 3)
out["config3"]["self_traverse_ms"] = round(ms_self, 3)
out["config3"]["self_contacts"] = ts.num_contacts
del tris, vols, bvh, tr, ts, p, d
torch.cuda.empty_cache()
# ---- the reference's published case (README.md:226-231): a 249,882-triangle mesh, 100,000 random rays ---------
# (the dragon itself is absent; the torus surrogate at the same triangle count stands in; A100: build 0.41 ms,
# traverse 1.14 ms, 1e5 rays 2.0 ms — other hardware, other mesh: context only)
tris_s = torch.from_numpy(torus_mesh(353, 354)).cuda()
vols_s = ibvh.bounding_volumes_from_triangles(tris_s)
ss = {"bvh": None, "t": None, "r": None}


def build_s():
    ss["bvh"] = ibvh.BVH(vols_s, cache=ss["bvh"])
    return ss["bvh"]


def trav_s():
    ss["t"] = ibvh.traverse(ss["bvh"], cache=ss["t"])
    return ss["t"]


ms_bs, _ = timed(build_s, 20)
ms_ts, tts = timed(trav_s, 20)
hv = vols_s.cpu().numpy()
lo, hi = hv[:, :3].min(0), hv[:, :3].max(0)
rng = np.random.default_rng(47)
ps = torch.from_numpy((lo + (hi - lo) * rng.random((100_000, 3))).astype(np.float32)).cuda().t()
ds = torch.from_numpy(rng.random((100_000, 3)).astype(np.float32)).cuda().t()


def rays_s():
    ss["r"] = ibvh.traverse_rays(ss["bvh"], ps, ds, cache=ss["r"])
    return ss["r"]


ms_rs, trs = timed(rays_s, 10)
out["readme_case_surrogate"] = {"triangles": int(tris_s.shape[0]), "build_ms": round(ms_bs, 4), "traverse_ms": round(ms_ts, 4),
                                "contacts": tts.num_contacts, "rays": 100_000, "traverse_rays_ms": round(ms_rs, 4),
                                "ray_hits": trs.num_contacts,
                                "build_plus_traverse_Mleaves_per_s": round(int(tris_s.shape[0]) / (ms_bs + ms_ts) / 1e3, 1)}
del tris_s, vols_s, ss, tts, trs, ps, ds
torch.cuda.empty_cache()
# ---- config 4 -------------------------------------------------------------------------------
n = 5_000_000
r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
a = ibvh.generate_spheres(n, 44, r0=r0)
b = ibvh.generate_spheres(n, 45, origin=(0.9, 0.0, 0.0), r0=r0)
ms_b, b1 = timed(lambda: ibvh.BVH(a))
b2 = ibvh.BVH(b)
st = {"t": None}


def pair():
    st["t"] = ibvh.traverse(b1, b2, cache=st["t"])
    return st["t"]


ms_pair, tp = timed(pair, 3)
out["config4"] = {"leaves_each": n, "build_ms_each": round(ms_b, 3), "pair_traverse_lvt_ms": round(ms_pair, 3),
                  "contacts": tp.num_contacts, "Mcontacts_per_s": round(tp.num_contacts / ms_pair / 1e3, 2),
                  "kernels_ms": kernels(pair)}
st["t"] = None
ms_bfs, tb = timed(lambda: ibvh.traverse(b1, b2, ibvh.BFSTraversal()), 2)
out["config4"]["pair_traverse_bfs_ms"] = round(ms_bfs, 3)
out["config4"]["bfs_checks"] = tb.num_checks
print(json.dumps(out))
