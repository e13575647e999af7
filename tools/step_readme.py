"""The published-size step in a loop, for rocprofv3: build (cache= chain) + LVT self-traverse + count read on the 249,882-leaf mesh.
usage: rocprofv3 --kernel-trace --stats -d gpurun_out/prof_readme -o r -- python3 tools/step_readme.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import implicitbvh_amd as ibvh  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
vols, _ = bench.readme_mesh_volumes(ibvh, torch)
b = t = None
for _ in range(20):
    b = ibvh.BVH(vols, cache=b)
    t = ibvh.traverse(b, cache=t)
    _ = t.num_contacts
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    b = ibvh.BVH(vols, cache=b)
    t = ibvh.traverse(b, cache=t)
    _ = t.num_contacts
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.4f} ms per step, {t.num_contacts} contacts, {len(b.leaves)} leaves")
