"""Per-launch durations of the BFS traversal's kernels (config 2: 1e6 spheres), from the library's launch profile.
usage: python tools/attic/bfs_levels.py [n]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from implicitbvh_amd.synthetic import sphere_radius_law
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
g = ibvh.BVH(ibvh.generate_spheres(n, 42, r0=sphere_radius_law(n)))
t = None
for _ in range(5):
    t = ibvh.traverse(g, ibvh.BFSTraversal(), cache=t)
    _ = t.num_contacts
torch.cuda.synchronize()
lib.call("ibvh_profile_enable", 1)
t = ibvh.traverse(g, ibvh.BFSTraversal(), cache=t)
_ = t.num_contacts
torch.cuda.synchronize()
cnt = C.c_int64(); lib.call("ibvh_profile_count", C.byref(cnt))
tot = 0
for i in range(cnt.value):
    name, ms = C.c_char_p(), C.c_float()
    lib.call("ibvh_profile_get", i, C.byref(name), C.byref(ms))
    tot += ms.value
    print(f"{i:3d} {ms.value*1e3:8.1f} us  {name.value.decode()[:90]}")
print("sum", round(tot * 1e3, 1), "us; contacts", t.num_contacts, "checks", t.num_checks, "start level", t.start_level1)
