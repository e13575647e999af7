"""Run ON THE GPU BOX: work counters of the LVT walks (ibvh_lvt_work_counters) next to the reference walk's (oracle)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import implicitbvh_amd as ibvh
import oracle_lib as orc
from implicitbvh_amd import abi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000
r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
v = ibvh.generate_spheres(n, 42, r0=r0)
b = ibvh.BVH(v)
host = orc.generate_spheres_f32(n, 42, r0=r0)
o = orc.build(host, abi.make_types())
print("self hip", ibvh.lvt_work_counters(b), "contacts", ibvh.traverse(b).num_contacts)
print("self ref", orc.lvt_test_counts(o, threads=8))
v2 = ibvh.generate_spheres(n, 45, origin=(0.9, 0, 0), r0=r0)
b2 = ibvh.BVH(v2)
o2 = orc.build(orc.generate_spheres_f32(n, 45, origin=(0.9, 0, 0), r0=r0), abi.make_types())
print("pair hip", ibvh.lvt_work_counters(b, b2), "contacts", ibvh.traverse(b, b2).num_contacts)
print("pair ref", orc.lvt_test_counts(o, o2, threads=8))
rng = np.random.default_rng(1); p = rng.random((20000, 3)).astype(np.float32); d = rng.random((20000, 3)).astype(np.float32)
print("rays hip", ibvh.lvt_work_counters(b, points=torch.from_numpy(p).cuda().t(), directions=torch.from_numpy(d).cuda().t()))
print("rays ref", orc.lvt_test_counts(o, points=p, directions=d, threads=8))
