#!/usr/bin/env python3
"""One rank of tests/test_gpu_dist_procs.py (run ON THE GPU BOX, one fresh OS process per rank, all ranks on the box's one GPU):
the PRODUCT's distributed driver — ibvh_dist_plan / ibvh_dist_exchange / ibvh_dist_cross_* of libibvh.so — over a collective
vtable whose callbacks stage through host memory into torch.distributed's gloo backend (device -> host -> gloo collective ->
device).  RCCL cannot put two ranks on one device; the driver does not care what is behind its vtable.
  mode "build N TOL": N leaves per rank; the concatenation of the ranks' slices == the single-device build of all leaves, byte for
                      byte; per-slice self contacts + cross-shard contacts == the single-device contact list as a set
                      (TOL = 0 forces the splitter refinement and the count-exchange branch of ibvh_dist_plan)
  mode "starved"    : every leaf has the same Morton code: one rank would receive nothing — EVERY rank must raise DomainError.
  mode "bad_args"   : one rank's arguments are not acceptable: it raises its own error, the others IBVH_ERR_PEER, nobody hangs."""
import datetime
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

mode = sys.argv[1]
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))  # (before anything touches the GPU)
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi  # noqa: E402
from implicitbvh_amd import dist as ibd  # noqa: E402


class GlooStagedComm:
    """the three collectives of ibvh_comm on DEVICE tensors, staged through host memory into gloo"""

    def __init__(self):
        self.rank, self.size = rank, world
        self.calls = {"all_reduce": 0, "all_gather": 0, "all_to_all": 0}

    def all_reduce(self, t, op):
        self.calls["all_reduce"] += 1
        h = t.cpu()
        dist.all_reduce(h, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op])
        t.copy_(h)
        return t

    def all_gather(self, t):
        self.calls["all_gather"] += 1
        h = t.cpu().contiguous()
        rows = [torch.empty_like(h) for _ in range(self.size)]
        dist.all_gather(rows, h)
        return torch.stack(rows).to(t.device)

    def all_to_all(self, send, send_counts, recv_counts):
        self.calls["all_to_all"] += 1
        h = send.cpu().contiguous()
        recv = torch.empty(int(sum(recv_counts)), dtype=h.dtype)
        dist.all_to_all_single(recv, h, output_split_sizes=[int(c) for c in recv_counts], input_split_sizes=[int(c) for c in send_counts])
        return recv.to(send.device)


comm = GlooStagedComm()
if mode == "starved":
    vols = torch.tensor([[0.5, 0.5, 0.5, 0.01]], dtype=torch.float32, device="cuda").repeat(1000, 1)
    builder = ibd.DistributedBuilder(comm)
    try:
        builder.build(vols)
    except abi.DomainError:
        print(f"ok rank {rank}: DomainError on this rank too", flush=True)
        dist.barrier()
        sys.exit(0)
    print(f"rank {rank}: the build did not stop", flush=True)
    sys.exit(1)

if mode == "bad_args":
    # round 6: ONE rank's arguments are not acceptable (a negative tolerance for the build; a negative cache_slots for the
    # completion): that rank reports its own error, every other rank IBVH_ERR_PEER — together, nobody is left in a collective
    n_rank = 20_000
    vols = ibvh.generate_spheres(n_rank, 46, first_index=rank * n_rank, r0=0.01)
    culprit = world - 1
    builder = ibd.DistributedBuilder(comm, tolerance=-1.0 if rank == culprit else 0.005)
    try:
        builder.build(vols)
        print(f"rank {rank}: the build did not stop", flush=True)
        sys.exit(1)
    except ValueError as e:
        assert rank == culprit, e
    except RuntimeError as e:
        assert rank != culprit and "another rank" in str(e), e
    builder = ibd.DistributedBuilder(comm)
    bvh = builder.build(vols)
    try:
        builder.cross_contacts(bvh, cache_slots=-1 if rank == culprit else None)
        print(f"rank {rank}: the completion did not stop", flush=True)
        sys.exit(1)
    except ValueError as e:
        assert rank == culprit, e
    except RuntimeError as e:
        assert rank != culprit and "another rank" in str(e), e
    cross = builder.cross_contacts(bvh)  # ... and the communicator is still usable
    torch.cuda.synchronize()
    print(f"ok rank {rank}: both calls stopped on every rank, then worked ({cross.shape[0]} cross contacts)", flush=True)
    dist.barrier()
    sys.exit(0)

n_rank, tol = int(sys.argv[2]), float(sys.argv[3])
n = n_rank * world
r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
vols = ibvh.generate_spheres(n_rank, 46, first_index=rank * n_rank, r0=r0)
builder = ibd.DistributedBuilder(comm, tolerance=tol)
bvh = builder.build(vols)
bvh2 = builder.build(vols, cache=bvh)  # the time-stepping shape: buffers reused
own = ibvh.traverse(bvh).contacts
cross = builder.cross_contacts(bvh)
torch.cuda.synchronize()
# two builds (one record exchange each) + the cross-shard completion: the counts, then ONE exchange of the selected leaves
# (round 6; world - 1 sequential rounds before); with tolerance 0 each build exchanges its counts as well
expect = 2 + 2 + (2 if tol == 0.0 else 0)
assert comm.calls["all_to_all"] == expect and comm.calls["all_gather"] >= 3, (comm.calls, expect)
lb = builder.last_cross
assert all(0 < c <= int(sizes_hint) for c in lb["leaves_received"]) if (sizes_hint := n) else True
if tol == 0.0:
    assert builder.last["levels_used"] > 12, builder.last["levels_used"]  # refined below the first 12-bit digit: counts were exchanged
# the single-device build of ALL leaves, made by this rank for itself
single = ibvh.BVH(ibvh.generate_spheres(n, 46, r0=r0))
sizes = torch.zeros(world, dtype=torch.int64)
sizes[rank] = len(bvh.leaves)
dist.all_reduce(sizes)
assert int(sizes.sum()) == n, sizes
lb = bvh.leaves.buf.numel() // len(bvh.leaves)
off = int(sizes[:rank].sum()) * lb
assert torch.equal(bvh.leaves.buf, single.leaves.buf[off:off + bvh.leaves.buf.numel()]), "this rank's slice differs from the single-device sorted sequence"
assert torch.equal(bvh2.leaves.buf, bvh.leaves.buf) and torch.equal(bvh2.nodes, bvh.nodes)
assert builder.last["extrema"].tolist() == single.extrema.cpu().numpy().tolist()
if tol > 0:
    assert abs(len(bvh.leaves) - n_rank) <= 0.02 * n_rank + 1, (len(bvh.leaves), n_rank)
# contacts: global indices everywhere; (min, max) order for the comparison
mine = torch.cat([own.long(), cross.long()]).cpu()
mine = torch.stack([mine.min(1).values, mine.max(1).values], 1)
counts = torch.zeros(world, dtype=torch.int64)
counts[rank] = mine.shape[0]
dist.all_reduce(counts)
ref = ibvh.traverse(single).contacts.long().cpu()
assert int(counts.sum()) == ref.shape[0], (counts.tolist(), ref.shape[0])
gathered = [torch.empty((int(c), 2), dtype=torch.int64) for c in counts]
dist.all_gather(gathered, mine) if len(set(counts.tolist())) == 1 else None
if len(set(counts.tolist())) != 1:  # (gloo's all_gather wants equal shapes: pad)
    m = int(counts.max())
    pad = torch.zeros((m, 2), dtype=torch.int64)
    pad[:mine.shape[0]] = mine
    rows = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(rows, pad)
    gathered = [rows[r][:int(counts[r])] for r in range(world)]
if rank == 0:
    allc = torch.cat(gathered)
    key = lambda c: torch.sort(c[:, 0] * (n + 1) + c[:, 1]).values  # noqa: E731
    ref2 = torch.stack([ref.min(1).values, ref.max(1).values], 1)
    assert torch.equal(key(allc), key(ref2)), "per-slice + cross-shard contacts != the single-device contact set"
print(f"ok rank {rank}: slice {len(bvh.leaves)} leaves, own {own.shape[0]} + cross {cross.shape[0]} contacts, partners {builder.last_cross['partners']}, "
      f"splitter bits {builder.last['levels_used']}", flush=True)
dist.barrier()
