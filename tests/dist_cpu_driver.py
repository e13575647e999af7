"""CPU stand-in for the distributed build's driver (csrc/ibvh_distdrv.hip needs a GPU): the same sequence of collectives
and the PRODUCT's splitter arithmetic (ibvh_splitter_search_* of libibvh, host-only code) around an engine that does the
per-rank data work on CPU (tests/test_dist_cpu.py: the oracle).  Test infrastructure: lets gloo processes and virtual
ranks exercise the N > 1 logic — splitters, send matrix, count exchange, global numbering, the "every rank stops together"
rule, cross-shard completion — without a GPU.  It is NOT the product's driver: that one needs a GPU and is run in real OS
processes by tests/test_gpu_dist_procs.py (gloo staged through host memory) and with virtual ranks by tests/test_gpu_parity.py;
what these CPU tests pin is the protocol and the library's host-side splitter arithmetic."""
import ctypes as C

import numpy as np
import torch

from implicitbvh_amd import abi, api, lib
from implicitbvh_amd import dist as ibd

DIGIT_BITS = 12


def find_splitters(engine, comm, keys, key_bits, n_global, tolerance, first_hist):
    L = lib.load()
    s = abi.SplitterSearch()
    abi.check(L.ibvh_splitter_search_init(C.byref(s), comm.size, key_bits, n_global, float(tolerance)), "ibvh_splitter_search_init")
    if comm.size == 1:
        return [], 0
    h = np.ascontiguousarray(first_hist, dtype=np.int64)
    abi.check(L.ibvh_splitter_search_step(C.byref(s), h.ctypes.data_as(C.POINTER(C.c_int64))), "ibvh_splitter_search_step")
    while not s.all_done:
        rows = [int(s.rows[j]) for j in range(s.num_rows)]
        hist = engine.histogram(keys, s.next_shift, s.next_bits, s.next_shift + s.next_bits, rows)
        comm.all_reduce(hist, "sum")
        h = np.ascontiguousarray(engine.to_host(hist), dtype=np.int64)
        abi.check(L.ibvh_splitter_search_step(C.byref(s), h.ctypes.data_as(C.POINTER(C.c_int64))), "ibvh_splitter_search_step")
    return [int(s.splitters[k]) for k in range(comm.size - 1)], int(s.decided)


# The cross-shard completion in its SIMPLEST form, for engines without a GPU: whole trees of touching slices change hands and the
# ordinary pair traversal runs against them.  The product (ibvh_dist_cross_*, csrc/ibvh_distdrv.hip) ships only the leaves whose
# box touches the receiver's boxes and builds a tree over them on arrival — fewer bytes, the same contact SET; the GPU tests
# (test_gpu_dist_procs.py, test_gpu_parity.py, test_gpu_dist.py) hold the product to the single-device list, this CPU version
# pins what that list is for sharded builds.
def cross_contacts(comm, eng, types, n_slice, bvh):
    """Cross-shard contact completion: contacts between leaves of DIFFERENT slices.

    Root boxes of all slices are all-gathered; for every pair of slices (r < s) whose root boxes touch, rank s
    copies its sorted leaves + nodes to rank r over xGMI and rank r runs the ordinary pair traversal
    (ibvh_traverse_pair_lvt_*) of its tree against the received one.  Returns this rank's share as an (m, 2)
    tensor of GLOBAL 1-based indices (index in own slice, index in the other slice).  The union over ranks of
    the per-slice self contacts and these pairs is the contact set of the whole cloud."""
    P, me = comm.size, comm.rank
    boxes = eng.tensor([[0.0] * 6] * P, torch.float64)
    boxes[me] = eng.root_box(bvh)
    comm.all_reduce(boxes, "sum")
    sizes = eng.tensor([0] * P, torch.int64)
    sizes[me] = n_slice
    comm.all_reduce(sizes, "sum")
    bx, sz = eng.to_host(boxes), eng.to_host(sizes).tolist()

    def touch(a, b):
        return bool(np.all(bx[a][3:] >= bx[b][:3]) and np.all(bx[a][:3] <= bx[b][3:]))
    out = []
    payload = None
    for d in range(1, P):  # round d: rank s sends to rank s - d (if their boxes touch)
        dst, src = me - d, me + d
        send_counts, recv_counts = [0] * P, [0] * P
        if dst >= 0 and touch(dst, me):
            if payload is None:
                payload = eng.export(bvh)
            send_counts[dst] = payload.numel()
        if src < P and touch(me, src):
            recv_counts[src] = eng.export_bytes(types, sz[src])
        send = payload if sum(send_counts) else eng.tensor([], torch.uint8)
        recv = comm.all_to_all(send, send_counts, recv_counts)
        if sum(recv_counts):
            other = eng.import_(types, sz[src], recv)
            out.append(eng.pair_contacts(bvh, other))
    return eng.cat(out) if out else eng.empty_contacts(types)


class CpuDistributedBuilder:
    def __init__(self, comm, engine, tolerance=0.005):
        self.comm, self.engine, self.tolerance, self.last = comm, engine, tolerance, {}

    def build(self, volumes, node_type=None, cache=None, options=None):
        eng, comm = self.engine, self.comm
        options = options or api.BVHOptions()
        node_type = node_type or api.BBox(torch.float32)
        kind, flt = (abi.BSPHERE if volumes.shape[1] == 4 else abi.BBOX), api._float_code(volumes.dtype)
        types = abi.make_types(kind, flt, node_type.kind, node_type.flt, options.index_code, options.morton_code)
        n_local = volumes.shape[0]
        fdt = abi.FLOAT_DTYPES[flt]
        fmax, fmin = float(np.finfo(fdt).max), float(np.finfo(fdt).tiny)
        # ONE all-reduce(MAX) of [-mins, maxs, one-hot leaf counts] (neutral elements of morton/utils.jl:29-40)
        vec = eng.tensor([-fmax] * 3 + [fmin] * 3 + [0.0] * comm.size, torch.float64)
        if n_local:
            e = eng.extrema(types, volumes).to(torch.float64)
            vec[:3] = -e[:3]
            vec[3:6] = e[3:]
        vec[6 + comm.rank] = float(n_local)
        if comm.size > 1:
            comm.all_reduce(vec, "max")
        ext = torch.cat([-vec[:3], vec[3:6]]).to(volumes.dtype)
        eng.expand(types, ext)
        keys = eng.keys(types, volumes, ext)
        key_bits = abi.MORTON_BITS[types.morton_type]
        bits0 = min(DIGIT_BITS, key_bits)
        shift0 = key_bits - bits0
        hist0 = eng.histogram(keys, shift0, bits0, 64, []).reshape(-1)
        allh = comm.all_gather(hist0) if comm.size > 1 else hist0.reshape(1, -1)
        H = eng.to_host(allh).astype(np.int64).reshape(comm.size, -1)
        counts = [int(round(c)) for c in eng.to_host(vec)[6:6 + comm.size]]
        ext_host = eng.to_host(ext).astype(fdt)
        base, n_global = int(sum(counts[:comm.rank])), int(sum(counts))
        if n_global < comm.size:
            raise abi.DomainError("fewer leaves than ranks")
        splitters, levels_used = find_splitters(eng, comm, keys, key_bits, n_global, self.tolerance, H.sum(0))
        send_matrix = None
        if comm.size > 1 and levels_used <= bits0:
            edges = [0] + [sp >> shift0 for sp in splitters] + [1 << bits0]
            cum = np.concatenate([np.zeros((comm.size, 1), np.int64), np.cumsum(H, axis=1)], axis=1)
            send_matrix = np.stack([cum[:, edges[r + 1]] - cum[:, edges[r]] for r in range(comm.size)], axis=1)  # [src, dst]
        known = send_matrix[comm.rank].tolist() if send_matrix is not None else None
        perm, send_counts = eng.partition(keys, splitters, comm.size, known)
        records, rec_bytes = eng.pack(types, volumes, keys, perm, base)
        if send_matrix is not None:
            recv_counts = send_matrix[:, comm.rank].tolist()
        elif comm.size > 1:
            rc = comm.all_to_all(eng.tensor(send_counts, torch.int64), [1] * comm.size, [1] * comm.size)
            recv_counts = eng.to_host(rc).tolist()
        else:
            recv_counts = list(send_counts)
        if comm.size > 1:
            recv = comm.all_to_all(records, [c * rec_bytes for c in send_counts], [c * rec_bytes for c in recv_counts])
        else:
            recv = records
        n_recv = int(sum(recv_counts))
        if send_matrix is not None:
            min_recv = int(send_matrix.sum(axis=0).min())
        elif comm.size > 1:
            flag = eng.tensor([n_recv], torch.int64)
            comm.all_reduce(flag, "min")
            min_recv = int(eng.to_host(flag)[0])
        else:
            min_recv = n_recv
        if min_recv < 1:
            raise abi.DomainError("a rank received no leaves (degenerate key distribution): every rank stops here")
        self.last = {"splitters": splitters, "send_counts": send_counts, "recv_counts": recv_counts, "base": base, "n_global": n_global,
                     "extrema": ext_host, "record_bytes": rec_bytes, "types": types, "n_slice": n_recv, "levels_used": levels_used}
        return eng.build_local(types, recv, n_recv, ext_host, node_type, options, cache)

    def cross_contacts(self, bvh):
        return cross_contacts(self.comm, self.engine, self.last["types"], self.last["n_slice"], bvh)
