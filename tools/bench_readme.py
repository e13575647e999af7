"""The reference's published workload alone (README.md:226-231): 249,882 triangle spheres, build / LVT self-traverse /
100,000 rays — bench.run_readme_250k without the rest of the bench.  usage: python tools/bench_readme.py [--cpu]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import lib  # noqa: E402

cpu = None
if "--cpu" in sys.argv:
    import oracle_lib as orc
    cpu = (orc, orc.load_native(), min(os.cpu_count() or 1, 32)) if hasattr(orc, "load_native") else None
lib.load()
res = bench.run_readme_250k(ibvh, lib, torch, cpu)
print(json.dumps(res, indent=1))
v, _ = bench.readme_mesh_volumes(ibvh, torch)
b = None
for i in range(6):
    b = ibvh.BVH(v, cache=b)
    torch.cuda.synchronize()
    print('chain build', i, 'hint', hex(int(ibvh.api._host_words().words[b._skew.slot])), 'asked levels/equalize', b._fast[1].sort_levels, b._fast[1].sort_equalize)
