#!/usr/bin/env python3
"""Run ON THE GPU BOX: one very large build + traversal (default 1.4e8 leaves: 29 tree levels, 64-bit queue entries)
checked on the device: codes ascending, .index a permutation, ties stable, records follow their index."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicitbvh_amd as ibvh
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 140_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
b = ibvh.BVH(v)
torch.cuda.synchronize(); t0 = time.perf_counter()
b = ibvh.BVH(v, cache=b)
torch.cuda.synchronize()
print("n", n, "levels", b.tree.levels, "build %.2f ms" % ((time.perf_counter() - t0) * 1e3), "hint", int(b._skew[0]))
m = b.leaves.morton
assert bool((m[1:] >= m[:-1]).all()), "codes not ascending"
idx = b.leaves.index
same = m[1:] == m[:-1]
assert bool((idx[1:][same] > idx[:-1][same]).all()), "ties out of input order"
del same
seen = torch.zeros(n + 1, dtype=torch.bool, device="cuda")
seen[idx.long()] = True
assert bool(seen[1:].all()), "not a permutation"
del seen
sel = torch.randint(0, n, (1_000_000,), device="cuda")
assert b.leaves.volume[sel].equal(v[idx[sel].long() - 1]), "records do not follow their index"
t = ibvh.traverse(b)
torch.cuda.synchronize(); t0 = time.perf_counter()
t = ibvh.traverse(b, cache=t)
nc = t.num_contacts
print("traverse %.2f ms" % ((time.perf_counter() - t0) * 1e3), "contacts", nc, "per leaf %.3f" % (nc / n))
c = t.contacts
assert bool((c[:, 0] < c[:, 1]).all())
s = torch.randint(0, nc, (2_000_000,), device="cuda")
a_, b_ = v[c[s, 0].long() - 1], v[c[s, 1].long() - 1]
d2 = ((a_[:, :3] - b_[:, :3]) ** 2).sum(1)
assert bool((d2 <= (a_[:, 3] + b_[:, 3]) ** 2 * (1 + 1e-5)).all()), "a reported pair does not touch"
print("ok")
