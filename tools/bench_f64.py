import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import implicitbvh_amd as ibvh
from implicitbvh_amd.synthetic import sphere_radius_law
n = 1_000_000
v64 = ibvh.generate_spheres(n, 42, r0=sphere_radius_law(n)).double()
for name, nt in (("bbox_f32_nodes", None), ("bbox_f64_nodes", ibvh.BBox(torch.float64))):
    b = t = None
    for _ in range(10):
        b = ibvh.BVH(v64, nt, cache=b); t = ibvh.traverse(b, cache=t); _ = t.num_contacts
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        b = ibvh.BVH(v64, nt, cache=b); t = ibvh.traverse(b, cache=t); _ = t.num_contacts
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / 50 * 1e3, 4), "ms per step", t.num_contacts)
    import bench
    from implicitbvh_amd import lib
    def step():
        global b, t
        b = ibvh.BVH(v64, nt, cache=b); t = ibvh.traverse(b, cache=t); return t
    tot = {}
    for _ in range(5):
        _, _, ks = bench._dominant(lib, torch, step)
        for k, v in ks.items(): tot[k] = tot.get(k, 0) + v / 5
    print("  ", {k: round(v, 4) for k, v in tot.items()})
