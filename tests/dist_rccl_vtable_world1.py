#!/usr/bin/env python3
"""Child process of tests/test_gpu_dist.py (run ON THE GPU BOX): the C-ABI collective vtable over RCCL — what a Julia host
binds (include/ibvh.h: ibvh_comm_from_rccl, ibvh_dist_plan, ibvh_dist_exchange) — with NO torch.distributed anywhere: an
ncclComm_t is created with RCCL's own C API (ncclGetUniqueId + ncclCommInitRank, world size 1: the one GPU of the box),
wrapped by ibvh_comm_from_rccl, its three entries are called directly on device buffers (identities at world size 1, but
through librccl.so: dlopen, symbol resolution, stream ordering), and the distributed build through it must equal the
single-device build bit for bit.
usage: dist_rccl_vtable_world1.py N"""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import abi, api, lib

n = int(sys.argv[1])
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


rccl = None
for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
    try:
        rccl = C.CDLL(name)
        break
    except OSError:
        pass
assert rccl is not None, "librccl.so not found"
uid = UniqueId()
rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0 and comm.value
L = lib.load()
vt = abi.Comm()
abi.check(L.ibvh_comm_from_rccl(comm, 0, 1, C.byref(vt)), "ibvh_comm_from_rccl")
stream = api._stream()
# the three collectives, straight through the function pointers
t = torch.arange(8, dtype=torch.float64, device="cuda")
assert vt.all_reduce(vt.ctx, t.data_ptr(), 8, abi.COMM_F64, abi.COMM_MAX, stream) == 0
i = torch.arange(5, dtype=torch.int64, device="cuda")
assert vt.all_reduce(vt.ctx, i.data_ptr(), 5, abi.COMM_I64, abi.COMM_SUM, stream) == 0
src = torch.arange(40, dtype=torch.uint8, device="cuda")
dst = torch.zeros(40, dtype=torch.uint8, device="cuda")
assert vt.all_gather(vt.ctx, src.data_ptr(), dst.data_ptr(), 40, stream) == 0
out = torch.zeros(24, dtype=torch.uint8, device="cuda")
cnt = (C.c_int64 * 1)(24)
assert vt.all_to_all_v(vt.ctx, src.data_ptr(), cnt, out.data_ptr(), cnt, stream) == 0
torch.cuda.synchronize()
assert t.tolist() == list(range(8)) and i.tolist() == list(range(5)) and dst.equal(src) and out.equal(src[:24])
# the distributed build through the vtable (raw C ABI, the sequence INTEGRATION.md gives a Julia host)
r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
vols = ibvh.generate_spheres(n, 46, r0=r0)
types = abi.make_types()
need = C.c_size_t()
abi.check(L.ibvh_dist_scratch_bytes(C.byref(types), n, 1, C.byref(need)), "ibvh_dist_scratch_bytes")
scratch = torch.empty(need.value, dtype=torch.uint8, device="cuda")
plan = abi.DistPlan()
abi.check(L.ibvh_dist_plan(C.byref(types), C.byref(vt), vols.data_ptr(), n, 0.005, scratch.data_ptr(), need.value, C.byref(plan), stream), "ibvh_dist_plan")
assert plan.n_slice == n and plan.n_global == n and plan.base == 0 and plan.size == 1
recv = torch.empty(n * plan.record_bytes, dtype=torch.uint8, device="cuda")
abi.check(L.ibvh_dist_exchange(C.byref(types), C.byref(vt), vols.data_ptr(), C.byref(plan), scratch.data_ptr(), need.value, recv.data_ptr(), stream),
          "ibvh_dist_exchange")
single = ibvh.BVH(vols)
opts = ibvh.BVHOptions(morton=ibvh.DefaultMortonAlgorithm(abi.MORTON_DTYPES[abi.U32], compute_extrema=False,
                                                           mins=tuple(plan.extrema[:3]), maxs=tuple(plan.extrema[3:])))
bvh = ibvh.BVH(ibvh.BoundingVolumes(types, n, recv), options=opts)
torch.cuda.synchronize()
assert single.extrema.double().cpu().tolist() == list(plan.extrema), "global extrema differ from the single-device build's"
assert bvh.leaves.buf.equal(single.leaves.buf) and bvh.nodes.equal(single.nodes), "dist through the RCCL vtable != single-device build"
print("rccl vtable world1", n, "ok", flush=True)
