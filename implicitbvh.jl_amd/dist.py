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

The collective layer is a small `comm` object so the same driver runs over torch.distributed (RCCL on
GPUs, gloo in CPU tests) or over in-process virtual ranks; the per-rank device work is an `engine`
(HipEngine = libibvh; tests inject an oracle-backed CPU engine to exercise the host logic without a GPU).
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
# per-rank device work
# ---------------------------------------------------------------------------------------------
class HipEngine:
    """libibvh on the current GPU."""

    def __init__(self):
        self.torch = api._require_gpu()
        self.device = "cuda"

    def tensor(self, data, dtype):
        return self.torch.tensor(data, dtype=dtype, device=self.device)

    def extrema(self, types, vols):
        torch = self.torch
        ext = torch.empty(6, dtype=vols.dtype, device=self.device)
        scratch = torch.empty(1 << 17, dtype=torch.uint8, device=self.device)
        lib.call("ibvh_extrema", C.byref(types), api._ptr(vols), 0, vols.shape[0], 0, api._ptr(ext), api._ptr(scratch),
                 scratch.numel(), api._stream())
        return ext

    def expand(self, types, ext):
        lib.call("ibvh_expand_extrema", types.leaf_float, api._ptr(ext), api._stream())
        return ext

    def pack_extrema(self, types, volumes, rank, nranks):
        """[-mins, maxs, one-hot leaf counts] as float64 on device: local extrema + ONE small kernel."""
        torch = self.torch
        n_local = volumes.shape[0]
        vec = torch.empty(6 + nranks, dtype=torch.float64, device=self.device)
        e = self.extrema(types, volumes) if n_local else None
        lib.call("ibvh_dist_pack_extrema", types.leaf_float, api._ptr(e) if n_local else None, 1 if n_local else 0, rank, nranks,
                 n_local, api._ptr(vec), api._stream())
        return vec

    def unpack_extrema(self, types, vec, dtype):
        """global extrema in the leaf float type, epsilon-expanded, from the reduced vector (one kernel)"""
        ext = self.torch.empty(6, dtype=dtype, device=self.device)
        lib.call("ibvh_dist_unpack_extrema", types.leaf_float, api._ptr(vec), api._ptr(ext), api._stream())
        return ext

    def keys(self, types, vols, ext):
        torch = self.torch
        kd = torch.int64 if types.morton_type == abi.U64 else torch.int32
        keys = torch.empty(vols.shape[0], dtype=kd, device=self.device)
        if vols.shape[0] == 0:  # a rank without leaves still takes part in every collective
            return keys
        lib.call("ibvh_morton_keys", C.byref(types), api._ptr(vols), 0, vols.shape[0], api._ptr(ext), api._ptr(keys),
                 api._stream())
        return keys

    def histogram(self, keys, shift, bits, prefix_shift, prefixes, raw=False):
        torch = self.torch
        rows = max(len(prefixes), 1)
        out = torch.empty((rows, 1 << bits), dtype=torch.int32, device=self.device)
        if keys.numel() == 0:
            out.zero_()
            return out if raw else out.to(torch.int64)
        arr = (C.c_uint64 * max(len(prefixes), 1))(*[int(p) for p in prefixes]) if prefixes else None
        lib.call("ibvh_key_histogram", keys.element_size(), api._ptr(keys), keys.numel(), shift, bits, prefix_shift, arr,
                 len(prefixes), api._ptr(out), api._stream())
        return out if raw else out.to(torch.int64)

    def partition(self, keys, splitters, nranks, known_counts=None):
        """Stable partition of the local leaves by destination rank: (perm, counts per rank), always inside the library
        (ibvh_dist_partition: destination kernel + ONE stable radix pass).  `known_counts`: this rank's row of the send
        matrix when the caller already derived it from the all-gathered histograms; otherwise the destination kernel
        counts as well and the row is read back (one small device -> host copy)."""
        torch = self.torch
        n = keys.numel()
        if nranks == 1:
            return None, [n]
        if nranks > 256:
            raise ValueError("the distributed build supports at most 256 ranks (ibvh_dist_partition)")
        if n == 0:
            return torch.empty(0, dtype=torch.int32, device=self.device), [0] * nranks
        perm = torch.empty(n, dtype=torch.int32, device=self.device)
        counts = None if known_counts is not None else torch.empty(nranks, dtype=torch.int64, device=self.device)
        need = C.c_size_t()
        lib.call("ibvh_dist_partition_scratch_bytes", n, C.byref(need))
        scratch = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        arr = (C.c_uint64 * max(len(splitters), 1))(*[int(sp) for sp in splitters])
        lib.call("ibvh_dist_partition", keys.element_size(), api._ptr(keys), n, arr, nranks, api._ptr(perm),
                 api._ptr(counts) if counts is not None else None, api._ptr(scratch), need.value, api._stream())
        return perm, (list(known_counts) if known_counts is not None else counts.cpu().tolist())

    def pack(self, types, vols, keys, perm, index_base):
        torch = self.torch
        lay = abi.Layout()
        lib.call("ibvh_layout_of", C.byref(types), C.byref(lay))
        out = torch.empty(vols.shape[0] * lay.leaf_bytes, dtype=torch.uint8, device=self.device)
        if vols.shape[0] == 0:
            return out, lay.leaf_bytes
        lib.call("ibvh_pack_records", C.byref(types), api._ptr(vols), api._ptr(keys), api._ptr(perm), int(index_base),
                 vols.shape[0], api._ptr(out), api._stream())
        return out, lay.leaf_bytes

    def build_local(self, types, records, n, ext_host, node_type, options, cache):
        fixed = api.DefaultMortonAlgorithm(options.morton.exemplar, compute_extrema=False,
                                           mins=tuple(float(v) for v in ext_host[:3]),
                                           maxs=tuple(float(v) for v in ext_host[3:]))
        opts = api.BVHOptions(index=options.index, morton=fixed, block_size=options.block_size)
        bv = api.BoundingVolumes(types, n, records)
        return api.BVH(bv, node_type, cache=cache, options=opts, _out_of_place=True)

    def to_host(self, t):
        return t.cpu().numpy()

    def event_pair(self):
        """two HIP events for timing a span of the current stream (the all-to-all: torch's RCCL backend makes its own
        stream wait for the current one and the current one wait for the collective)"""
        return self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)

    # ---- cross-shard completion ------------------------------------------------------------------
    def root_box(self, bvh):
        """(lo, up) of the slice as 6 float64: the root node, or the single leaf's box."""
        torch = self.torch
        if bvh.nodes.shape[0] > 0:
            v = bvh.nodes[0].to(torch.float64)
        else:
            v = bvh.leaves.volume[0].to(torch.float64)
        if v.numel() == 4:  # sphere -> its box (conservative in float64)
            v = torch.cat([v[:3] - v[3], v[:3] + v[3]])
        return v

    def export(self, bvh):
        """leaves ‖ nodes as one byte tensor for the peer copy."""
        torch = self.torch
        return torch.cat([bvh.leaves.buf.view(torch.uint8), bvh.nodes.contiguous().view(torch.uint8).reshape(-1)])

    def import_(self, types, n, buf):
        torch = self.torch
        lay = abi.Layout()
        lib.call("ibvh_layout_of", C.byref(types), C.byref(lay))
        lb = n * lay.leaf_bytes
        tree = api.ImplicitTree(n)
        nn = tree.real_nodes - tree.real_leaves
        leaves = buf[:lb].clone()
        ndt = api._torch_float(types.node_float)
        nodes = buf[lb:lb + nn * lay.node_bytes].clone().view(ndt).reshape(nn, abi.volume_width(types.node_kind))
        return api.BVH.from_buffers(types, n, leaves, nodes)

    def export_bytes(self, types, n):
        lay = abi.Layout()
        lib.call("ibvh_layout_of", C.byref(types), C.byref(lay))
        tree = api.ImplicitTree(n)
        return n * lay.leaf_bytes + (tree.real_nodes - tree.real_leaves) * lay.node_bytes

    def pair_contacts(self, bvh_a, bvh_b):
        return api.traverse(bvh_a, bvh_b).contacts

    def empty_contacts(self, types):
        return self.torch.empty((0, 2), dtype=api._torch_index(types.index_type), device=self.device)

    def cat(self, ts):
        return self.torch.cat(ts)


# ---------------------------------------------------------------------------------------------
# splitter search (identical arithmetic on every rank)
# ---------------------------------------------------------------------------------------------
DIGIT_BITS = 12


def find_splitters(engine, comm, keys, key_bits, n_global, tolerance=0.005, first_hist=None):
    """Keys k_1 <= ... <= k_{P-1}: rank r receives the keys in [k_r, k_{r+1}).

    Digit histograms are refined from the top of the key, 12 bits per level (one all-reduce(SUM) of <= 15 x 4096
    counters each).  After a level, splitter s sits on the digit whose cumulative count first exceeds the balanced
    target s*N/P; the digits below it are decided, and the splitter may stop there (k_s = prefix << remaining bits)
    once the bucket it landed in holds at most tolerance*N/P keys — the imbalance it can cause.  tolerance = 0
    refines to full key resolution: #(keys < k_s) <= s*N/P < #(keys <= k_s).  `first_hist`: the already reduced
    level-0 histogram (host array), when the caller has folded it into an earlier collective."""
    P = comm.size
    if P == 1:
        return [], 0
    targets = [s * n_global // P for s in range(1, P)]
    allowed = tolerance * n_global / P
    prefix = [0] * (P - 1)   # bits decided so far, as a value
    below = [0] * (P - 1)    # global number of keys strictly below the decided prefix range
    done = [False] * (P - 1)
    final = [0] * (P - 1)
    decided = 0
    while decided < key_bits and not all(done):
        bits = min(DIGIT_BITS, key_bits - decided)
        shift = key_bits - decided - bits
        if decided == 0:
            rows, row_of = [], [0] * (P - 1)
            if first_hist is not None:
                h = np.asarray(first_hist, dtype=np.int64).reshape(1, -1)
            else:
                hist = engine.histogram(keys, shift, bits, 64, [])
                comm.all_reduce(hist, "sum")
                h = engine.to_host(hist).astype(np.int64)
        else:
            rows = sorted({prefix[s] for s in range(P - 1) if not done[s]})
            row_of = [rows.index(p) if not done[i] else 0 for i, p in enumerate(prefix)]
            hist = engine.histogram(keys, shift, bits, shift + bits, rows)
            comm.all_reduce(hist, "sum")
            h = engine.to_host(hist).astype(np.int64)
        cum = np.cumsum(h, axis=1)
        for s in range(P - 1):
            if done[s]:
                continue
            r = row_of[s]
            rem = targets[s] - below[s]
            d = int(np.searchsorted(cum[r], rem, side="right"))  # first digit with inclusive count > rem
            d = min(d, (1 << bits) - 1)
            below[s] += int(cum[r][d - 1]) if d > 0 else 0
            prefix[s] = (prefix[s] << bits) | d
            if shift == 0 or int(h[r][d]) <= allowed:
                done[s] = True
                final[s] = prefix[s] << shift
        decided += bits
    return final, decided


# ---------------------------------------------------------------------------------------------
# the driver
# ---------------------------------------------------------------------------------------------
class DistributedBuilder:
    """builder = DistributedBuilder(comm_or_group); bvh = builder.build(local_volumes, node_type, cache=..., options=...)

    `local_volumes`: this rank's (n_local, 4|6) volumes; global leaf g = (sum of lower ranks' counts) + local
    position; the returned BVH's leaves carry .index = g + 1."""

    def __init__(self, comm=None, engine=None, tolerance=0.005):
        if comm is None or not hasattr(comm, "all_reduce"):
            comm = TorchComm(comm)
        self.comm = comm
        self.engine = engine or HipEngine()
        self.tolerance = tolerance  # allowed imbalance per splitter, as a fraction of N/P (0 = exact)
        self.last = {}
        self.time_exchange = False  # exchange_stats(): bracket the all-to-all with HIP events

    def build(self, volumes, node_type=None, cache=None, options=None):
        eng, comm = self.engine, self.comm
        options = options or api.BVHOptions()
        torch = api._torch()
        node_type = node_type or api.BBox(torch.float32)
        kind, flt = (abi.BSPHERE if volumes.shape[1] == 4 else abi.BBOX), api._float_code(volumes.dtype)
        types = abi.make_types(kind, flt, node_type.kind, node_type.flt, options.index_code, options.morton_code)
        if not abi.combo_supported(types):
            raise ValueError("unsupported leaf / node type combination")
        n_local = volumes.shape[0]
        fdt = abi.FLOAT_DTYPES[flt]
        fmax, fmin = float(np.finfo(fdt).max), float(np.finfo(fdt).tiny)
        # 1. ONE all-reduce(MAX) of a float64 vector [-mins, maxs, one-hot leaf counts]: the global centre AABB
        #    (float -> double -> float is exact, and min(x) = -max(-x) exactly) and every rank's leaf count
        #    (global numbering) in a single collective.  Neutral elements of the reference's reduces
        #    (morton/utils.jl:29-40): floatmax for the minima, floatmin for the maxima.
        if hasattr(eng, "pack_extrema"):
            vec = eng.pack_extrema(types, volumes, comm.rank, comm.size)
        else:  # engines without the fused kernels (the CPU test engine)
            vec = eng.tensor([-fmax] * 3 + [fmin] * 3 + [0.0] * comm.size, torch.float64)
            if n_local:
                e = eng.extrema(types, volumes).to(torch.float64)
                vec[:3] = -e[:3]
                vec[3:6] = e[3:]
            vec[6 + comm.rank] = float(n_local)
        if comm.size > 1:
            comm.all_reduce(vec, "max")
        if hasattr(eng, "unpack_extrema"):
            ext = eng.unpack_extrema(types, vec, volumes.dtype)
        else:
            ext = torch.cat([-vec[:3], vec[3:6]]).to(volumes.dtype)
            eng.expand(types, ext)
        # 2. keys + this rank's first splitter histogram; the histograms are ALL-GATHERED (P x 4096 counters), so every
        #    rank knows the whole send matrix as soon as the splitters sit on first-level bucket boundaries (the
        #    normal case) and no count exchange is needed.  ONE device->host copy (raw bytes of the three pieces)
        #    brings back everything the host needs.
        keys = eng.keys(types, volumes, ext)
        key_bits = abi.MORTON_BITS[types.morton_type]
        bits0 = min(DIGIT_BITS, key_bits)
        shift0 = key_bits - bits0
        try:
            hist0 = eng.histogram(keys, shift0, bits0, 64, [], raw=True).reshape(-1)
        except TypeError:
            hist0 = eng.histogram(keys, shift0, bits0, 64, []).reshape(-1)
        allh = comm.all_gather(hist0) if comm.size > 1 else hist0.reshape(1, -1)
        allh = allh.contiguous()
        blob = eng.to_host(torch.cat([vec.view(torch.uint8), ext.contiguous().view(torch.uint8), allh.view(torch.uint8).reshape(-1)]))
        nv, ne = 8 * (6 + comm.size), 6 * ext.element_size()
        host_vec = np.frombuffer(blob[:nv].tobytes(), dtype=np.float64)
        counts = [int(round(c)) for c in host_vec[6:6 + comm.size]]
        ext_host = np.frombuffer(blob[nv:nv + ne].tobytes(), dtype=fdt).copy()
        hdt = np.int32 if allh.element_size() == 4 else np.int64
        H = np.frombuffer(blob[nv + ne:].tobytes(), dtype=hdt).astype(np.int64).reshape(comm.size, -1)
        base, n_global = int(sum(counts[:comm.rank])), int(sum(counts))
        if n_global < comm.size:
            raise abi.DomainError("fewer leaves than ranks")
        # 3. splitters, partition, pack, exchange
        splitters, levels_used = find_splitters(eng, comm, keys, key_bits, n_global, self.tolerance, first_hist=H.sum(0))
        send_matrix = None
        if comm.size > 1 and levels_used <= bits0:
            # splitters are multiples of 2^shift0: destination of a key depends on its first digit only
            edges = [0] + [sp >> shift0 for sp in splitters] + [1 << bits0]
            cum = np.concatenate([np.zeros((comm.size, 1), np.int64), np.cumsum(H, axis=1)], axis=1)
            send_matrix = np.stack([cum[:, edges[r + 1]] - cum[:, edges[r]] for r in range(comm.size)], axis=1)  # [src, dst]
        known = send_matrix[comm.rank].tolist() if send_matrix is not None else None
        perm, send_counts = eng.partition(keys, splitters, comm.size, known)
        records, rec_bytes = eng.pack(types, volumes, keys, perm, base)
        if send_matrix is not None:
            recv_counts = send_matrix[:, comm.rank].tolist()
        elif comm.size > 1:
            sc = eng.tensor(send_counts, torch.int64)
            rc = comm.all_to_all(sc, [1] * comm.size, [1] * comm.size)
            recv_counts = eng.to_host(rc).tolist()
        else:
            recv_counts = list(send_counts)
        events = eng.event_pair() if self.time_exchange and hasattr(eng, "event_pair") else None
        if events:
            events[0].record()
        if comm.size > 1:
            recv = comm.all_to_all(records, [c * rec_bytes for c in send_counts], [c * rec_bytes for c in recv_counts])
        else:
            recv = records
        if events:
            events[1].record()
        n_recv = int(sum(recv_counts))
        # A rank left without leaves (all its keys' neighbours are duplicates of one splitter key, or the cloud is
        # heavily clustered) cannot build a tree.  EVERY rank must learn that and raise together: a rank that
        # carried on alone would hang in the next collective until the RCCL timeout.
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
        self.last = {"splitters": splitters, "send_counts": send_counts, "recv_counts": recv_counts, "base": base,
                     "n_global": n_global, "extrema": ext_host, "record_bytes": rec_bytes, "exchange_events": events}
        # 4. local build over the received slice
        bvh = eng.build_local(types, recv, n_recv, ext_host, node_type, options, cache)
        self.last["types"] = types
        self.last["n_slice"] = n_recv
        return bvh

    def exchange_stats(self, volumes, node_type=None, options=None, repeats=3):
        """What this rank's share of the distributed sort exchange costs: `repeats` builds with the all-to-all bracketed
        by HIP events.  Returns {"rank", "exchange_ms" (mean), "bytes_sent", "bytes_received" (both WITHOUT the part
        that stays on this GPU), "peers"}: with 7 xGMI links of ~153 GB/s per GPU the exchange is per-link bound, so
        bytes_sent / peers / exchange_ms against 153 GB/s is the figure to watch on a real node."""
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

    def cross_contacts(self, bvh):
        """Cross-shard contact completion: contacts between leaves of DIFFERENT slices.

        Root boxes of all slices are all-gathered; for every pair of slices (r < s) whose root boxes touch, rank s
        copies its sorted leaves + nodes to rank r over xGMI and rank r runs the ordinary pair traversal
        (ibvh_traverse_pair_lvt_*) of its tree against the received one.  Returns this rank's share as an (m, 2)
        tensor of GLOBAL 1-based indices (index in own slice, index in the other slice).  The union over ranks of
        the per-slice self contacts and these pairs is the contact set of the whole cloud."""
        eng, comm = self.engine, self.comm
        torch = api._torch()
        types = self.last["types"]
        P, me = comm.size, comm.rank
        boxes = eng.tensor([[0.0] * 6] * P, torch.float64)
        boxes[me] = eng.root_box(bvh)
        comm.all_reduce(boxes, "sum")
        sizes = eng.tensor([0] * P, torch.int64)
        sizes[me] = self.last["n_slice"]
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


# ---------------------------------------------------------------------------------------------
# in-process virtual ranks (single-GPU emulation of P shards; used by tests and for debugging)
# ---------------------------------------------------------------------------------------------
class ThreadWorld:
    def __init__(self, size):
        import threading
        self.size = size
        self.barrier = threading.Barrier(size)
        self.slots = [None] * size


class ThreadComm:
    """Collectives between `size` Python threads of one process (one virtual rank each)."""

    def __init__(self, world, rank):
        self.world, self.rank, self.size = world, rank, world.size

    def all_reduce(self, t, op):
        import torch
        w = self.world
        w.slots[self.rank] = t.clone()
        w.barrier.wait()
        stack = torch.stack(w.slots)
        res = {"min": lambda s: s.min(0).values, "max": lambda s: s.max(0).values, "sum": lambda s: s.sum(0)}[op](stack)
        w.barrier.wait()
        t.copy_(res)
        return t

    def all_gather(self, t):
        import torch
        w = self.world
        w.slots[self.rank] = t.clone()
        w.barrier.wait()
        out = torch.stack(w.slots)
        w.barrier.wait()
        return out

    def all_to_all(self, send, send_counts, recv_counts):
        import torch
        w = self.world
        w.slots[self.rank] = (send, [int(c) for c in send_counts])
        w.barrier.wait()
        pieces = []
        for src in range(self.size):
            s, sc = w.slots[src]
            off = sum(sc[:self.rank])
            pieces.append(s[off:off + sc[self.rank]])
            assert sc[self.rank] == int(recv_counts[src])
        out = torch.cat(pieces) if pieces else send[:0]
        w.barrier.wait()
        return out


def run_virtual_ranks(size, fn):
    """Run fn(comm) on `size` virtual ranks (threads); returns the list of results in rank order."""
    import threading
    world = ThreadWorld(size)
    results, errors = [None] * size, []

    def work(r):
        try:
            results[r] = fn(ThreadComm(world, r))
        except BaseException as e:  # noqa: BLE001 - re-raised below
            errors.append(e)
            world.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results
