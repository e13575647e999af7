#!/usr/bin/env python3
"""Run ON THE GPU BOX: builds of skewed inputs (clusters, a surface mesh, duplicates), each checked against the CPU
oracle where it is small enough, else for sortedness / permutation / stability; each case in its own process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = f'''
import sys, math
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {ROOT!r} + "/tests")
import numpy as np, torch, implicitbvh_amd as ibvh
from implicitbvh_amd import abi
case, n = sys.argv[1], int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(11)
if case == "clusters":
    c = torch.rand((8, 3), generator=g, device="cuda")
    w = torch.randint(0, 8, (n,), generator=g, device="cuda")
    v = torch.empty((n, 4), device="cuda"); v[:, :3] = c[w] + 0.004 * torch.randn((n, 3), generator=g, device="cuda"); v[:, 3] = 1e-4
elif case == "onecluster":
    v = torch.empty((n, 4), device="cuda"); v[:, :3] = 0.5 + 0.001 * torch.randn((n, 3), generator=g, device="cuda"); v[:, 3] = 1e-4
    v[0, :3] = 100.0  # a far outlier collapses the grid
elif case == "dups":
    base = torch.rand((1000, 4), generator=g, device="cuda")
    v = base[torch.randint(0, 1000, (n,), generator=g, device="cuda")].contiguous()
elif case == "uniform":
    v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
elif case == "mesh":
    from implicitbvh_amd.synthetic import torus_mesh
    tris = torch.from_numpy(torus_mesh()).cuda()
    v = ibvh.bounding_volumes_from_triangles(tris); n = v.shape[0]
b = ibvh.BVH(v)
torch.cuda.synchronize()
m = b.leaves.morton
assert bool((m[1:] >= m[:-1]).all()), "not sorted"
idx = b.leaves.index.cpu()
assert idx.sort().values.equal(torch.arange(1, n + 1, dtype=idx.dtype)), "not a permutation"
same = (m[1:] == m[:-1])
assert bool((idx[1:][same] > idx[:-1][same]).all()), "ties out of input order"
sel = torch.randint(0, n, (50000,), device="cuda")
assert b.leaves.volume[sel].equal(v[(b.leaves.index[sel].long() - 1)]), "volumes do not follow their index"
if n <= 400000:
    import oracle_lib as orc
    o = orc.build(v.cpu().numpy(), abi.make_types())
    assert b.leaves.to_numpy().tobytes() == o.leaves.tobytes(), "leaves differ from the oracle"
    assert b.nodes.cpu().numpy().tobytes() == o.nodes.tobytes(), "nodes differ from the oracle"
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): b = ibvh.BVH(v, cache=b)
torch.cuda.synchronize()
print("ok", case, n, "build %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), "levels used", int(b._skew[0]))
from implicitbvh_amd import lib
sys.path.insert(0, {ROOT!r})
from bench import collect_profile
lib.call("ibvh_profile_enable", 1)
b = ibvh.BVH(v, cache=b); torch.cuda.synchronize()
p = collect_profile(lib)
lib.call("ibvh_profile_enable", 0)
print("   ", {{k: (round(x[0] * 1e3, 1), x[1]) for k, x in p.items()}}, "sum %.1f us" % sum(x[0] * 1e3 for x in p.values()))
'''
for spec in sys.argv[1:]:
    case, n = spec.split(":")
    r = subprocess.run([sys.executable, "-c", code, case, n], capture_output=True, text=True, timeout=600)
    print(spec, "->", (r.stdout.strip() or "FAILED rc=%d %s" % (r.returncode, r.stderr.strip()[-400:])), flush=True)
