#!/usr/bin/env python3
"""Run ON THE GPU BOX: 1e8 leaves in 8 tight clusters, the build through equalised cells (forced) against the plain route: leaves
and nodes byte for byte (one-off check of the route at the largest single-device size; ~3 s)."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, math
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
n = 100_000_000
g = torch.Generator(device="cuda").manual_seed(5)
c = torch.rand((8, 3), generator=g, device="cuda")
v = torch.empty((n, 4), dtype=torch.float32, device="cuda")
which = torch.randint(0, 8, (n,), generator=g, device="cuda")
v[:, :3] = c[which] + 0.004 * torch.randn((n, 3), generator=g, device="cuda")
v[:, 3] = 1e-5
del which
lib.set_tuning("msd_equalize", -1)
plain = ibvh.BVH(v)
torch.cuda.synchronize()
ref_l = plain.leaves.buf.clone(); ref_n = plain.nodes.clone()
del plain; torch.cuda.empty_cache()
lib.set_tuning("msd_equalize", 1)
eq = ibvh.BVH(v)
torch.cuda.synchronize()
print("1e8 clustered: equalised == plain:", bool(torch.equal(eq.leaves.buf, ref_l)), bool(torch.equal(eq.nodes, ref_n)), "hint %#x" % int(ibvh.api._host_words().words[eq._skew.slot]))
