import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import implicitbvh_amd as ibvh
from implicitbvh_amd.synthetic import random_rays
vols, _ = bench.readme_mesh_volumes(ibvh, torch)
b = ibvh.BVH(vols)
hv = vols[:, :3]; lo, hi = hv.min(0).values.cpu().numpy(), hv.max(0).values.cpu().numpy()
for nr in (100_000, 1_000_000):
    p_host, d_host = random_rays(nr, lo, hi, seed=43)
    p, d = torch.from_numpy(p_host).cuda().t(), torch.from_numpy(d_host).cuda().t()
    r = None
    for _ in range(5):
        r = ibvh.traverse_rays(b, p, d, cache=r); _ = r.num_contacts
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        r = ibvh.traverse_rays(b, p, d, cache=r); _ = r.num_contacts
    torch.cuda.synchronize()
    print(nr, "rays:", round((time.perf_counter() - t0) / 30 * 1e3, 4), "ms", r.num_contacts, "hits")
