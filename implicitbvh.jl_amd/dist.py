"""Multi-GPU BVH build: leaves sharded over the GPUs of one node, one process per GPU.

The reference is single-device; this is the MI355X-native scaling path BASELINE.json's north star asks
for.  Only the BUILD communicates; traversal stays per GPU:

  1. per-GPU centre extrema (ibvh_extrema, unexpanded)  ->  ONE RCCL all-reduce(MAX) of [-mins, maxs, one-hot
     leaf counts] in float64: min/max are exact, so the global AABB — and after the same epsilon expansion
     the Morton codes — are bit-identical to the single-device build (morton/utils.jl:1-72);
  2. per-GPU Morton keys (ibvh_morton_keys);
  3. distributed radix sort: splitter keys found by refining 12-bit digit histograms from the top of the
     key (ibvh_key_histogram, one all-reduce(SUM) of <= 15 x 4096 counters per level; normally ONE level:
     refinement stops once a splitter's bucket is lighter than 0.5 % of a shard), so that rank r
     receives the keys in [k_r, k_{r+1}); stable partition of the local leaves by destination
     (one pass of the radix sort), pack into BoundingVolume records whose .index is the GLOBAL 1-based
     leaf number (ibvh_pack_records), ONE all-to-all of the records over xGMI;
  4. ordinary local build (ibvh_build, already_wrapped, fixed global extrema): stable LSB radix sort of
     the received records — they arrive grouped by source rank in source order, so ties keep global
     input order — gather, bottom-up merge.

Result: rank r holds the r-th contiguous slice of the globally stable-sorted leaf sequence (concatenating
the ranks' leaves gives exactly the single-device sorted array) and an implicit tree over its slice.
Contacts between leaves of different slices are not found by the per-GPU self-traversal.

Round 4: the driver itself lives BEHIND the C ABI (include/ibvh.h "multi-GPU build: the driver": ibvh_dist_plan,
ibvh_dist_exchange, ibvh_splitter_search_*, ibvh_comm / ibvh_comm_from_rccl; csrc/ibvh_distdrv.hip), so that a Julia host
can run it with nothing but ccall.  This module is the binding: it hands the library a collective vtable whose entries call
back into a small `comm` object (torch.distributed: RCCL on GPUs; in-process virtual ranks in tests), sizes the record
array from the plan and runs the ordinary local build.  Round 5: the cross-shard contact completion is behind the boundary
as well (ibvh_dist_cross_plan / _exchange / _count / _write over the same vtable); `cross_contacts` below sizes two buffers
from the plan and calls them.  Nothing in this module is an algorithm any more.
"""
import ctypes as C

import numpy as np

from . import abi, lib
from . import api


# ---------------------------------------------------------------------------------------------
# collectives
# ---------------------------------------------------------------------------------------------
class TorchComm:
    """torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def all_reduce(self, t, op):
        ops = {"min": self.dist.ReduceOp.MIN, "max": self.dist.ReduceOp.MAX, "sum": self.dist.ReduceOp.SUM}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
        return t

    def all_gather(self, t):
        """1-D tensor (same length on every rank) -> (size, len) tensor, row r from rank r."""
        import torch
        rows = [torch.empty_like(t) for _ in range(self.size)]
        self.dist.all_gather(rows, t.contiguous(), group=self.group)  # list form: works on gloo and RCCL alike
        return torch.stack(rows)

    def all_to_all(self, send, send_counts, recv_counts):
        """send: 1-D tensor partitioned by destination (send_counts elements each) -> received 1-D tensor."""
        import torch
        recv = torch.empty(int(sum(recv_counts)), dtype=send.dtype, device=send.device)
        self.dist.all_to_all_single(recv, send, output_split_sizes=[int(c) for c in recv_counts],
                                    input_split_sizes=[int(c) for c in send_counts], group=self.group)
        return recv


# ---------------------------------------------------------------------------------------------
# the collective vtable handed to the library (include/ibvh.h, ibvh_comm)
# ---------------------------------------------------------------------------------------------
class _DeviceBuffer:
    """A device pointer as something torch.as_tensor accepts without copying (__cuda_array_interface__)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class CommVtable:
    """ibvh_comm over a `comm` object with tensor-level all_reduce / all_gather / all_to_all (TorchComm, or the tests' virtual
    ranks).  The callbacks run on the calling thread with torch's current stream — the stream the library call was given."""

    def __init__(self, comm):
        self.comm = comm
        torch = api._torch()
        self._keep = []  # tensors a callback hands to an asynchronous collective stay alive until the next build

        def tensor(ptr, count, typestr):
            return torch.as_tensor(_DeviceBuffer(ptr, count, typestr), device="cuda")

        def all_reduce(ctx, buf, count, dtype, op, stream):
            try:
                t = tensor(buf, count, {abi.COMM_F64: "<f8", abi.COMM_I64: "<i8", abi.COMM_I32: "<i4"}[dtype])
                comm.all_reduce(t, {abi.COMM_MAX: "max", abi.COMM_SUM: "sum", abi.COMM_MIN: "min"}[op])
                self._keep.append(t)
                return 0
            except BaseException as e:  # noqa: BLE001 - an exception must not unwind through the C frames
                self.error = e
                return abi.ERR_HIP

        def all_gather(ctx, send, recv, nbytes, stream):
            try:
                s = tensor(send, nbytes, "|u1")
                r = tensor(recv, nbytes * comm.size, "|u1")
                r.copy_(comm.all_gather(s).reshape(-1))
                self._keep += [s, r]
                return 0
            except BaseException as e:  # noqa: BLE001
                self.error = e
                return abi.ERR_HIP

        def all_to_all_v(ctx, send, send_bytes, recv, recv_bytes, stream):
            try:
                sc = [int(send_bytes[p]) for p in range(comm.size)]
                rc = [int(recv_bytes[p]) for p in range(comm.size)]
                s = tensor(send, max(sum(sc), 1), "|u1")[:sum(sc)]
                r = tensor(recv, max(sum(rc), 1), "|u1")[:sum(rc)]
                out = comm.all_to_all(s, sc, rc)
                if out.data_ptr() != r.data_ptr():
                    r.copy_(out)
                self._keep += [s, r, out]
                return 0
            except BaseException as e:  # noqa: BLE001
                self.error = e
                return abi.ERR_HIP

        self.error = None
        self._cbs = (abi.COMM_ALL_REDUCE(all_reduce), abi.COMM_ALL_GATHER(all_gather), abi.COMM_ALL_TO_ALL_V(all_to_all_v))
        self.struct = abi.Comm(None, comm.rank, comm.size, *self._cbs)

    def begin(self):
        self._keep.clear()
        self.error = None

    def check(self, what, status):
        if self.error is not None:
            e, self.error = self.error, None
            raise e
        abi.check(status, what)


# ---------------------------------------------------------------------------------------------
# the binding
# ---------------------------------------------------------------------------------------------
class DistributedBuilder:
    """builder = DistributedBuilder(comm_or_group); bvh = builder.build(local_volumes, node_type, cache=..., options=...)

    `local_volumes`: this rank's (n_local, 4|6) volumes on the GPU; global leaf g = (sum of lower ranks' counts) + local
    position; the returned BVH's leaves carry .index = g + 1.  Everything up to the received records happens inside the
    library (ibvh_dist_plan, ibvh_dist_exchange); the local build is the ordinary BVH(...) over them."""

    def __init__(self, comm=None, tolerance=0.005):
        if comm is None or not hasattr(comm, "all_reduce"):
            comm = TorchComm(comm)
        self.comm = comm
        self.vtable = CommVtable(comm)
        self.tolerance = tolerance  # allowed imbalance per splitter, as a fraction of N/P (0 = exact)
        self.last = {}
        self.time_exchange = False  # exchange_stats(): bracket the all-to-all with HIP events
        self.time_phases = False    # bench.py --gpus N: synchronise between the phases of build() and keep their wall times in last["phases_ms"]
        self._scratch = None

    def build(self, volumes, node_type=None, cache=None, options=None):
        comm, vt = self.comm, self.vtable
        if comm.size > abi.DIST_MAX_RANKS:
            raise ValueError("the distributed build supports at most 256 ranks")
        options = options or api.BVHOptions()
        torch = api._torch()
        node_type = node_type or api.BBox(torch.float32)
        kind, flt = (abi.BSPHERE if volumes.shape[1] == 4 else abi.BBOX), api._float_code(volumes.dtype)
        types = abi.make_types(kind, flt, node_type.kind, node_type.flt, options.index_code, options.morton_code)
        if not abi.combo_supported(types):
            raise ValueError("unsupported leaf / node type combination")
        volumes = volumes.contiguous()
        n_local = volumes.shape[0]
        need = C.c_size_t()
        lib.call("ibvh_dist_scratch_bytes", C.byref(types), n_local, comm.size, C.byref(need))
        if self._scratch is None or self._scratch.numel() < need.value:
            self._scratch = torch.empty(need.value, dtype=torch.uint8, device="cuda")
        sp, sn = api._ptr(self._scratch), self._scratch.numel()
        plan = abi.DistPlan()
        vt.begin()
        phases, clock = {}, None
        if self.time_phases:
            import time
            torch.cuda.synchronize()
            clock = time.perf_counter

            def lap(name, t0):
                torch.cuda.synchronize()
                phases[name] = round((clock() - t0) * 1e3, 4)
                return clock()
            t_phase = clock()
        vt.check("ibvh_dist_plan", lib.load().ibvh_dist_plan(C.byref(types), C.byref(vt.struct), api._ptr(volumes), n_local,
                                                             float(self.tolerance), sp, sn, C.byref(plan), api._stream()))
        if clock:
            t_phase = lap("plan (extrema all-reduce, keys, histogram all-gather, splitters, partition)", t_phase)
        n_recv = int(plan.n_slice)
        lay = abi.Layout()
        lib.call("ibvh_layout_of", C.byref(types), C.byref(lay))
        recv = torch.empty(n_recv * lay.leaf_bytes, dtype=torch.uint8, device="cuda")
        events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if self.time_exchange else None
        if events:
            events[0].record()
        vt.check("ibvh_dist_exchange", lib.load().ibvh_dist_exchange(C.byref(types), C.byref(vt.struct), api._ptr(volumes), C.byref(plan),
                                                                     sp, sn, api._ptr(recv), api._stream()))
        if events:
            events[1].record()
        if clock:
            t_phase = lap("exchange (pack + all_to_all_v of the records)", t_phase)
        P = comm.size
        ext_host = np.array(list(plan.extrema), dtype=abi.FLOAT_DTYPES[flt])
        self.last = {"splitters": [int(plan.splitters[k]) for k in range(P - 1)], "send_counts": [int(plan.send_counts[r]) for r in range(P)],
                     "recv_counts": [int(plan.recv_counts[r]) for r in range(P)], "base": int(plan.base), "n_global": int(plan.n_global),
                     "extrema": ext_host, "record_bytes": int(plan.record_bytes), "exchange_events": events, "types": types,
                     "n_slice": n_recv, "levels_used": int(plan.levels_used)}
        # the ordinary local build over the received slice: pre-wrapped records, fixed global extrema, out of place
        fixed = api.DefaultMortonAlgorithm(options.morton.exemplar, compute_extrema=False,
                                           mins=tuple(float(v) for v in ext_host[:3]), maxs=tuple(float(v) for v in ext_host[3:]))
        opts = api.BVHOptions(index=options.index, morton=fixed, block_size=options.block_size)
        bv = api.BoundingVolumes(types, n_recv, recv)
        out = api.BVH(bv, node_type, cache=cache, options=opts, _out_of_place=True)
        if clock:
            lap("local build over the received slice", t_phase)
            self.last["phases_ms"] = phases
        return out

    def exchange_stats(self, volumes, node_type=None, options=None, repeats=3):
        """What this rank's share of the distributed sort exchange costs: `repeats` builds with the exchange (pack +
        all-to-all) bracketed by HIP events.  Returns {"rank", "exchange_ms" (mean), "bytes_sent", "bytes_received" (both
        WITHOUT the part that stays on this GPU), "peers"}: with 7 xGMI links of ~153 GB/s per GPU the exchange is per-link
        bound, so bytes_sent / peers / exchange_ms against 153 GB/s is the figure to watch on a real node."""
        self.time_exchange = True
        try:
            ms, cache = [], None
            for _ in range(max(1, repeats)):
                cache = self.build(volumes, node_type, cache=cache, options=options)
                ev = self.last.get("exchange_events")
                if ev:
                    ev[1].synchronize()
                    ms.append(ev[0].elapsed_time(ev[1]))
        finally:
            self.time_exchange = False
        me, rb = self.comm.rank, self.last["record_bytes"]
        sent = sum(c for r, c in enumerate(self.last["send_counts"]) if r != me) * rb
        received = sum(c for r, c in enumerate(self.last["recv_counts"]) if r != me) * rb
        return {"rank": me, "exchange_ms": round(sum(ms) / len(ms), 4) if ms else None, "bytes_sent": int(sent),
                "bytes_received": int(received), "peers": self.comm.size - 1}

    def cross_contacts(self, bvh, cache_slots=None):
        """Cross-shard contact completion (include/ibvh.h "Cross-shard contact completion"): this rank's
        share of the contacts between leaves of DIFFERENT slices as an (m, 2) tensor of GLOBAL 1-based indices (index in own
        slice, index in the other slice).  The union over ranks of the per-slice self contacts and these pairs is the contact
        set of the whole cloud.  Collective: every rank calls it."""
        torch = api._torch()
        vt, comm = self.vtable, self.comm
        L = lib.load()
        s = bvh.struct()
        k = api.LVT_CACHE_SLOTS if cache_slots is None else int(cache_slots)
        small = torch.empty(abi.dist_cross_scratch(comm.size), dtype=torch.uint8, device="cuda")
        plan = abi.DistCrossPlan()
        vt.begin()
        phases, t_phase = {}, None
        if self.time_phases:  # (bench.py: synchronise between the four calls and keep their wall times in last_cross["phases_ms"])
            import time
            torch.cuda.synchronize()
            t_phase = time.perf_counter()

        def lap(name):
            nonlocal t_phase
            if t_phase is not None:
                torch.cuda.synchronize()
                now = time.perf_counter()
                phases[name] = round((now - t_phase) * 1e3, 4)
                t_phase = now
        vt.check("ibvh_dist_cross_plan", L.ibvh_dist_cross_plan(C.byref(vt.struct), C.byref(s), k, api._ptr(small), small.numel(), C.byref(plan),
                                                                api._stream()))
        lap("cross plan (describe, all-gather, filter, all_to_all of the counts)")
        exp = torch.empty(max(int(plan.export_bytes), 1), dtype=torch.uint8, device="cuda")
        imp = torch.empty(max(int(plan.import_bytes), 1), dtype=torch.uint8, device="cuda")
        vt.check("ibvh_dist_cross_exchange", L.ibvh_dist_cross_exchange(C.byref(vt.struct), C.byref(s), C.byref(plan), api._ptr(exp), api._ptr(imp),
                                                                        api._ptr(small), small.numel(), api._stream()))
        lap("cross exchange (pack + all_to_all_v of the touching leaves)")
        scratch = torch.empty(max(int(plan.scratch_bytes), 1), dtype=torch.uint8, device="cuda")
        totals = (C.c_int64 * abi.DIST_MAX_RANKS)()
        total = C.c_int64()
        vt.check("ibvh_dist_cross_count", L.ibvh_dist_cross_count(C.byref(s), C.byref(plan), api._ptr(imp), api._ptr(scratch), scratch.numel(), totals,
                                                                  C.byref(total), api._stream()))
        lap("cross count (trees over the imported sets + pair traversals, counting)")
        out = torch.empty((int(total.value), 2), dtype=api._torch_index(bvh.types.index_type), device="cuda")
        if total.value > 0:  # (imported leaves without a single contact among them: nothing to write — as the Julia binding)
            vt.check("ibvh_dist_cross_write", L.ibvh_dist_cross_write(C.byref(s), C.byref(plan), api._ptr(imp), api._ptr(scratch), scratch.numel(),
                                                                      totals, api._ptr(out), api._stream()))
        lap("cross write")
        lay = abi.Layout()
        lib.call("ibvh_layout_of", C.byref(bvh.types), C.byref(lay))
        self.last_cross = {"partners": [int(plan.recv_rank[i]) for i in range(plan.n_recv)],
                           "leaves_received": [int(plan.recv_leaves[i]) for i in range(plan.n_recv)],
                           "bytes_received": int(sum(plan.recv_leaves[i] for i in range(plan.n_recv)) * lay.leaf_bytes),
                           "bytes_sent": int(sum(plan.send_leaves[r] for r in range(comm.size)) * lay.leaf_bytes),
                           "import_bytes": int(plan.import_bytes), "pairs": [int(totals[i]) for i in range(plan.n_recv)]}
        if phases:
            self.last_cross["phases_ms"] = phases
        return out
