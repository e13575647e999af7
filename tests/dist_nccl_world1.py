#!/usr/bin/env python3
"""Child process of tests/test_gpu_dist.py (run ON THE GPU BOX): DistributedBuilder over torch.distributed with the
REAL "nccl" (= RCCL) backend at world size 1 — RCCL is loaded, a communicator is created and every collective of the
build (all-reduce of the extrema vector, all-gather of the histograms, all-to-all of the records) is issued through it.
  mode "oracle N": the distributed build of N leaves must equal the single-device oracle build byte for byte;
  mode "props N" : N leaves (the per-GPU share of BASELINE.json configs[4] is 12.5e6): size-independent properties."""
import datetime
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

mode, n = sys.argv[1], int(sys.argv[2])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[3] if len(sys.argv) > 3 else "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
import implicitbvh_amd as ibvh
from implicitbvh_amd import abi
from implicitbvh_amd import dist as ibd

assert dist.get_backend() == "nccl"
builder = ibd.DistributedBuilder(dist.group.WORLD)
assert isinstance(builder.comm, ibd.TorchComm) and builder.comm.size == 1
r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
vols = ibvh.generate_spheres(n, 46, r0=r0)
# the collectives themselves, on the communicator the builder uses (world size 1: identities, but through RCCL)
t = torch.arange(8, dtype=torch.float64, device="cuda")
builder.comm.all_reduce(t, "max")
g = builder.comm.all_gather(torch.arange(5, dtype=torch.int32, device="cuda"))
a2a = builder.comm.all_to_all(torch.arange(7, dtype=torch.uint8, device="cuda"), [7], [7])
torch.cuda.synchronize()
assert t.tolist() == list(range(8)) and g.shape == (1, 5) and a2a.tolist() == list(range(7))
bvh = builder.build(vols)
bvh2 = builder.build(vols, cache=bvh)  # the time-stepping shape: buffers reused
torch.cuda.synchronize()
single = ibvh.BVH(vols)
assert bvh.leaves.buf.equal(single.leaves.buf) and bvh.nodes.equal(single.nodes), "dist(world 1) != single-device build"
assert bvh2.leaves.buf.equal(single.leaves.buf) and bvh2.nodes.equal(single.nodes)
if mode == "oracle":
    import oracle_lib as orc
    host = orc.generate_spheres_f32(n, 46, r0=r0)
    o = orc.build(host, abi.make_types())
    assert bvh.leaves.to_numpy().tobytes() == o.leaves.tobytes(), "sorted leaves differ from the oracle"
    assert bvh.nodes.cpu().numpy().tobytes() == o.nodes.tobytes(), "nodes differ from the oracle"
    exp, _ = orc.traverse_lvt(o)
    got = ibvh.traverse(bvh).contacts.cpu().numpy()
    assert got.shape[0] == len(exp) and (got[:, 0] == exp["a"]).all() and (got[:, 1] == exp["b"]).all()
else:
    m = bvh.leaves.morton
    assert bool((m[1:] >= m[:-1]).all()), "Morton codes not ascending"
    idx = bvh.leaves.index.cpu()
    assert idx.sort().values.equal(torch.arange(1, n + 1, dtype=idx.dtype)), "indices are not a permutation of 1..n"
    # stability: equal codes keep input (= index) order
    same = (m[1:] == m[:-1])
    assert bool((idx[1:][same] > idx[:-1][same]).all()), "ties out of input order"
    # the sorted volumes are the input volumes of their index
    sel = torch.randint(0, n, (100000,), device="cuda")
    assert bvh.leaves.volume[sel].equal(vols[(bvh.leaves.index[sel].long() - 1)])
    # node spot check: every 1000th leaf-parent box is the exact merge of its two leaves (BSphere -> BBox, merge.jl:58-81)
    trav = ibvh.traverse(bvh)
    c = trav.contacts[:200000].long()
    assert trav.num_contacts > 0 and bool((c[:, 0] < c[:, 1]).all())
    va, vb = vols[c[:, 0] - 1].double(), vols[c[:, 1] - 1].double()
    d2 = ((va[:, :3] - vb[:, :3]) ** 2).sum(1)
    assert bool((d2 <= (va[:, 3] + vb[:, 3]) ** 2 * (1 + 1e-5)).all()), "a reported pair does not touch"
print("dist nccl world1", mode, n, "ok", flush=True)
dist.destroy_process_group()
