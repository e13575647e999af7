"""Host-side mirror of ImplicitBVH.jl's hot-path API over libibvh (HIP, gfx950).

Same names, argument meaning and error behaviour as the reference:
    BVH(bounding_volumes, node_type; built_level, cache, options)      src/build.jl:198-271
    traverse(bvh[, bvh2], alg; start_level..., narrow, cache, options)  src/traverse/traverse.jl:210-233
    traverse_rays(bvh, points, directions, alg; ...)                    src/raytrace/raytrace.jl:71-80
    BVHOptions, DefaultMortonAlgorithm, LVTTraversal, BFSTraversal, BVHTraversal, ImplicitTree,
    default_start_level, memory_index, level_indices, isvirtual.
Device memory is torch tensors on the current CUDA(HIP) device; this module allocates buffers, fills
the POD descriptors of include/ibvh.h and calls the C entry points.  Nothing is computed on the host
and there is no CPU fallback.

Python renderings of Julia types:
    BSphere{T} arrays  -> float tensor of shape (n, 4): x, y, z, r
    BBox{T} arrays     -> float tensor of shape (n, 6): lo, up
    node_type          -> BSphere(dtype) / BBox(dtype) tokens, e.g. BBox(torch.float32)
    Vector{BoundingVolume{V,I,M}} -> BoundingVolumes (byte buffer with the Julia record layout)
    Vector{IndexPair{I}}          -> integer tensor of shape (n, 2)
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import abi, lib

__all__ = [
    "load_obj_triangles", "BSphere", "BBox", "BoundingVolumes", "BVHOptions", "DefaultMortonAlgorithm", "ImplicitTree", "BVH",
    "BVHTraversal", "LVTTraversal", "BFSTraversal", "traverse", "traverse_rays", "default_start_level",
    "memory_index", "level_indices", "isvirtual", "bounding_volumes_from_triangles", "generate_spheres",
    "NARROW_MORTON_LT", "NARROW_INDEX_LT", "NARROW_RAY_ORIGIN_OUTSIDE", "LeafBatch", "lvt_work_counters",
]

NARROW_MORTON_LT = abi.NARROW_MORTON_LT
NARROW_INDEX_LT = abi.NARROW_INDEX_LT
NARROW_RAY_ORIGIN_OUTSIDE = abi.NARROW_RAY_ORIGIN_OUTSIDE


def _torch():
    import torch
    return torch


_gpu_checked = False


def _require_gpu():
    global _gpu_checked
    torch = _torch()
    if _gpu_checked:  # (a GPU does not go away; torch.cuda.is_available() costs 2 us a call on the time-stepping path)
        return torch
    if not torch.cuda.is_available():
        raise RuntimeError("implicitbvh_amd needs an AMD GPU (MI355X / gfx950); there is no CPU fallback")
    lib.load()
    _gpu_checked = True
    return torch


_raw_stream = None  # torch._C._cuda_getCurrentRawStream when this torch has it: 0.3 us instead of ~9 us per call


def _stream():
    """torch's current stream on the current device as a void* for the C ABI.  This sits on the host's critical path of
    a time-stepping loop that reads the contact count every step (the GPU idles until the next build's first launch
    arrives), hence the raw accessor instead of torch.cuda.current_stream()."""
    global _raw_stream
    torch = _torch()
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return C.c_void_p(_raw_stream(torch._C._cuda_getDevice()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else 0)


def _float_code(dtype):
    torch = _torch()
    if dtype in (torch.float32, np.float32, "float32", float.__name__ + "32"):
        return abi.F32
    if dtype in (torch.float64, np.float64, "float64", float):
        return abi.F64
    raise ValueError(f"unsupported float type {dtype} (Float32 and Float64 are instantiated)")


def _torch_float(code):
    torch = _torch()
    return torch.float32 if code == abi.F32 else torch.float64


def _index_code(x):
    torch = _torch()
    if x in (torch.int32, np.int32, "int32") or isinstance(x, np.int32):
        return abi.I32
    if x in (torch.int64, np.int64, "int64", int) or isinstance(x, (np.int64, int)):
        return abi.I64
    raise ValueError(f"unsupported index type {x} (Int32 and Int64 are instantiated)")


def _torch_index(code):
    torch = _torch()
    return torch.int32 if code == abi.I32 else torch.int64


_morton_memo = {}


def _morton_code(x):
    if not isinstance(x, (type, str, np.dtype)):  # (instances of a Morton type: rare, not memoised)
        return _morton_code_slow(x)
    code = _morton_memo.get(x)
    if code is None:
        code = _morton_memo[x] = _morton_code_slow(x)
    return code


def _morton_code_slow(x):
    for code, dt in abi.MORTON_DTYPES.items():
        if x is dt or x == dt or isinstance(x, dt) or x == np.dtype(dt).name:
            return code
    raise ValueError(f"unsupported Morton type {x} (UInt16, UInt32, UInt64)")


# ---------------------------------------------------------------------------------------------
# type tokens
# ---------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class _VolumeType:
    kind: int
    flt: int

    def __repr__(self):
        return f"{'BSphere' if self.kind == abi.BSPHERE else 'BBox'}{{{'Float32' if self.flt == abi.F32 else 'Float64'}}}"


def BSphere(dtype=None):
    """Type token BSphere{T} (bsphere.jl:26-29)."""
    return _VolumeType(abi.BSPHERE, _float_code(dtype if dtype is not None else _torch().float32))


def BBox(dtype=None):
    """Type token BBox{T} (bbox.jl:35-38)."""
    return _VolumeType(abi.BBOX, _float_code(dtype if dtype is not None else _torch().float32))


@dataclass(frozen=True)
class DefaultMortonAlgorithm:
    """morton/default.jl:21-40.  `exemplar` is a numpy unsigned type (np.uint16/32/64)."""
    exemplar: type = np.uint32
    compute_extrema: bool = True
    mins: tuple = (float("nan"),) * 3
    maxs: tuple = (float("nan"),) * 3

    def __post_init__(self):
        _morton_code(self.exemplar)


@dataclass(frozen=True)
class BVHOptions:
    """utils.jl:34-93.  CPU threading knobs are kept for signature parity; only `index`, `morton` and
    `block_size` influence the GPU path (block_size is advisory: kernels pick their own geometry)."""
    index: object = np.int32
    morton: DefaultMortonAlgorithm = field(default_factory=DefaultMortonAlgorithm)
    num_threads: int = 1
    min_mortons_per_thread: int = 100
    min_sorts_per_thread: int = 100
    min_boundings_per_thread: int = 100
    min_traversals_per_thread: int = 100
    block_size: int = 256

    def __post_init__(self):
        for name in ("num_threads", "min_mortons_per_thread", "min_sorts_per_thread", "min_boundings_per_thread",
                     "min_traversals_per_thread", "block_size"):
            if not getattr(self, name) > 0:  # @argcheck ... > 0 (utils.jl:73-78)
                raise ValueError(f"BVHOptions: {name} > 0 must hold")
        _index_code(self.index)

    @property
    def index_code(self):
        return _index_code(self.index)

    @property
    def morton_code(self):
        return _morton_code(self.morton.exemplar)


class LVTTraversal:
    """Leaf-vs-tree traversal (traverse/leaf_vs_tree/leaf_vs_tree.jl:1), the default."""


class BFSTraversal:
    """Breadth-first BVTT traversal (traverse/breadth_first/breadth_first.jl:1)."""


# ---------------------------------------------------------------------------------------------
# ImplicitTree
# ---------------------------------------------------------------------------------------------
class ImplicitTree:
    """implicit_tree.jl:52-90."""

    def __init__(self, num_leaves):
        self._t = abi.Tree()
        lib.call("ibvh_tree_shape", int(num_leaves), C.byref(self._t))

    levels = property(lambda s: s._t.levels)
    real_leaves = property(lambda s: s._t.real_leaves)
    real_nodes = property(lambda s: s._t.real_nodes)
    virtual_leaves = property(lambda s: s._t.virtual_leaves)
    virtual_nodes = property(lambda s: s._t.virtual_nodes)

    def __repr__(self):
        return f"ImplicitTree(levels: {self.levels}, real_leaves: {self.real_leaves})"


def memory_index(tree, implicit_index):
    out = C.c_int64()
    try:
        lib.call("ibvh_memory_index", C.byref(tree._t), int(implicit_index), C.byref(out))
    except ValueError as e:
        raise IndexError(str(e))  # BoundsError (implicit_tree.jl:131-135)
    return out.value


def level_indices(tree, level):
    a, b = C.c_int64(), C.c_int64()
    try:
        lib.call("ibvh_level_indices", C.byref(tree._t), int(level), C.byref(a), C.byref(b))
    except ValueError as e:
        raise IndexError(str(e))
    return a.value, b.value


def isvirtual(tree, implicit_index):
    out = C.c_int32()
    try:
        lib.call("ibvh_isvirtual", C.byref(tree._t), int(implicit_index), C.byref(out))
    except ValueError as e:
        raise IndexError(str(e))
    return bool(out.value)


# Per-(types, n) constants of a build / traversal (tree shape, record layout, scratch sizes): pure functions of their key,
# asked of the library once.  A time-stepping caller rebuilds the same shape every step; on the step's critical path
# (the host reads the contact count, then enqueues the next build) every avoided ctypes round trip counts.
_shape_memo = {}


def _memo(key, make):
    v = _shape_memo.get(key)
    if v is None:
        if len(_shape_memo) > 256:
            _shape_memo.clear()
        v = _shape_memo[key] = make()
    return v


def _layout_of(types):
    def make():
        lay = abi.Layout()
        lib.call("ibvh_layout_of", C.byref(types), C.byref(lay))
        return lay
    return _memo(("layout", types.key()), make)


# ---------------------------------------------------------------------------------------------
# BoundingVolumes: Vector{BoundingVolume{V,I,M}} in device memory
# ---------------------------------------------------------------------------------------------
class BoundingVolumes:
    """Device array of BoundingVolume{V,I,M} records (bounding_volumes.jl:55-59) with the Julia layout."""

    def __init__(self, types, n, buf=None):
        torch = _torch()
        self.types = types
        self.n = int(n)
        lay = _layout_of(types)
        self.layout = lay
        if buf is None:
            buf = torch.zeros(self.n * lay.leaf_bytes, dtype=torch.uint8, device="cuda")
        self.buf = buf

    def __len__(self):
        return self.n

    @classmethod
    def wrap(cls, volumes, indices, options=None):
        """Manually wrapped volumes with user indices (build.jl:128-153); morton = 0."""
        torch = _require_gpu()
        options = options or BVHOptions()
        kind, flt = _volume_kind(volumes)
        types = abi.make_types(kind, flt, abi.BBOX, abi.F32, options.index_code, options.morton_code)
        out = cls(types, volumes.shape[0])
        out.volume.copy_(volumes)
        out.index.copy_(torch.as_tensor(indices, device="cuda").to(_torch_index(types.index_type)))
        return out

    def _strided(self, dtype, offset, width):
        torch = _torch()
        esz = torch.empty((), dtype=dtype).element_size()
        lb = self.layout.leaf_bytes
        assert lb % esz == 0 and offset % esz == 0
        flat = self.buf.view(dtype)
        return flat.as_strided((self.n, width), (lb // esz, 1), offset // esz)

    @property
    def volume(self):
        """(n, 4|6) strided float view of the .volume fields."""
        return self._strided(_torch_float(self.types.leaf_float), 0, abi.volume_width(self.types.leaf_kind))

    @property
    def index(self):
        return self._strided(_torch_index(self.types.index_type), self.layout.index_off, 1)[:, 0]

    @property
    def morton(self):
        """Morton codes as int64 (torch has no unsigned 32/64-bit arithmetic); values are exact for U16/U32
        and the two's-complement image for U64 (bit 63 is never set: 63 significant bits)."""
        torch = _torch()
        m = self.to_numpy()["morton"].astype(np.int64)
        return torch.from_numpy(m)

    @property
    def morton_device(self):
        """Morton codes as an int64 DEVICE tensor (a strided view widened on the fly; same value convention as .morton)."""
        torch = _torch()
        mb = {abi.U16: (torch.int16, 0xffff), abi.U32: (torch.int32, 0xffffffff), abi.U64: (torch.int64, None)}[self.types.morton_type]
        m = self._strided(mb[0], self.layout.morton_off, 1)[:, 0].to(torch.int64)
        return m if mb[1] is None else m & mb[1]

    def to_numpy(self):
        """Host copy as a numpy structured array (volume, index, morton)."""
        dt = abi.leaf_dtype(self.types)
        assert dt.itemsize == self.layout.leaf_bytes
        return self.buf.cpu().numpy().view(dt)


def _volume_kind(volumes):
    if volumes.dim() != 2 or volumes.shape[1] not in (4, 6):
        raise ValueError("bounding volumes must be an (n, 4) BSphere or (n, 6) BBox float tensor")
    if not volumes.is_cuda:
        raise ValueError("bounding volumes must live on the GPU (device='cuda')")
    return (abi.BSPHERE if volumes.shape[1] == 4 else abi.BBOX), _float_code(volumes.dtype)


# ---------------------------------------------------------------------------------------------
# words the GPU writes for the host (traversal totals, skew hints)
# ---------------------------------------------------------------------------------------------
class _HostWords:
    """ONE block of mapped pinned host memory for the whole process, allocated once and never freed, handed out word by
    word, round robin.  The GPU writes into it (the total of an enqueued traversal: `total_host` of include/ibvh.h; the
    skew hint of a build: `skew_flag`) and the host reads it WITHOUT a stream synchronisation or a device-to-host copy.
    Because the block is never returned to torch's caching host allocator, a kernel still in flight when its BVH /
    traversal object dies can only ever write into memory this module owns.  A word is recycled after SLOTS later
    requests; a reader holding an older generation of the slot is told so instead of being given another call's value."""
    SLOTS = 8192       # totals: [0, SLOTS)
    HINT_SLOTS = 1024  # skew hints: [SLOTS, SLOTS + HINT_SLOTS) — a ring of their own, so that a build still in flight
    #                    can at worst overwrite another chain's HINT (speed only), never a traversal's total
    PENDING = -(1 << 62)  # what the host stores before the launch; the GPU overwrites it

    def __init__(self):
        torch = _torch()
        self.tensor = torch.zeros(self.SLOTS + self.HINT_SLOTS, dtype=torch.int64).pin_memory()
        self.words = self.tensor.numpy()  # the same memory
        self.base = self.tensor.data_ptr()
        self.generation = [0] * (self.SLOTS + self.HINT_SLOTS)
        self.next = 0
        self.next_hint = 0
        self.devices_this_lap = set()  # devices whose streams may hold a scan kernel that writes a total slot of this lap
        import threading
        self.lock = threading.Lock()

    def take(self, initial, hint=False):
        with self.lock:
            if hint:
                slot = self.SLOTS + self.next_hint
                self.next_hint = (self.next_hint + 1) % self.HINT_SLOTS
            else:
                slot = self.next
                self.next = (self.next + 1) % self.SLOTS
                if self.next == 0:
                    # the ring wraps: a scan kernel of the previous lap that is STILL queued would overwrite the new
                    # owner's PENDING word with its own total.  One synchronisation per SLOTS traversals retires
                    # every such kernel before its slot is handed out again — of EVERY device this process drives (ADVICE
                    # r4: a traversal queued on another device is otherwise not retired), and under the lock on purpose:
                    # a thread that took a slot of the new lap meanwhile could launch behind a stale kernel of the old lap
                    # on its own stream and read that kernel's total as its own.  Cost: one device drain per 8,192 traversals.
                    # Only the devices that were handed a slot during the lap that just ended (in a one-process-per-GPU
                    # job every GPU is visible, and synchronising all of them creates a context — hundreds of MB — on each).
                    torch = _torch()
                    for d in sorted(self.devices_this_lap):
                        torch.cuda.synchronize(d)
                    self.devices_this_lap = set()
                self.devices_this_lap.add(_torch().cuda.current_device())
            self.generation[slot] += 1
            gen = self.generation[slot]
        self.words[slot] = initial
        return slot, gen

    def ptr(self, slot):
        return self.base + 8 * slot


class _HintWord:
    """A build chain's skew hint: one word of the _HostWords hint ring; word[0] reads / writes the value."""

    def __init__(self, initial):
        self.slot, self.gen = _host_words().take(initial, hint=True)
        self.holdoff = 0  # rebuilds for which the chain does not ask for equalised cells (apply)

    def valid(self):
        return _host_words().generation[self.slot] == self.gen

    def ptr(self):
        return _host_words().ptr(self.slot)

    def __getitem__(self, i):
        return int(_host_words().words[self.slot]) & 0xff  # extra levels the last build would have used (include/ibvh.h, skew_flag)

    def occupancy(self):
        """Fullest coarse cell of the last build in 1/128 of what one finish workgroup sorts (second byte of the word)."""
        return (int(_host_words().words[self.slot]) >> 8) & 0xff

    def __setitem__(self, i, v):
        _host_words().words[self.slot] = int(v)

    def equalize(self):
        """Bit 16: the last build ran with equalised cells and the plain grid would still have had a crowded cell."""
        return (int(_host_words().words[self.slot]) >> 16) & 1

    def apply(self, d):
        """sort_levels / sort_equalize of a rebuild from what the chain's previous build left in the word (read ONCE)."""
        d.sort_levels, d.sort_equalize, self.holdoff = sort_hint_rule(int(_host_words().words[self.slot]), d.n, self.holdoff)


def sort_hint_rule(word, n, holdoff):
    """(sort_levels, sort_equalize, holdoff') of a rebuild of n leaves from the hint word the previous build of its chain left
    (include/ibvh.h, ibvh_build_desc.skew_flag: low byte = extra partition levels that build would have used, second byte =
    its fullest cell in 1/128 of what a finish workgroup sorts, bit 16 = it ran with equalised cells and the plain grid would
    still have been crowded, bit 17 = equalising did not help, bit 18 = it ran with equalised cells) and the chain's counter
    `holdoff`: > 0 = rebuilds left on the plain grid after bit 17; -1 = PROBATION: the previous request went back to the plain
    grid on an equalised build's estimate (bit 16 = 0), with a spare level as insurance; <= -2 = STICKY: the probation build was
    crowded after all, so the chain stays with equalised cells for EQ_STICKY rebuilds whatever bit 16 says (round 6: a mesh
    whose fullest grid cell sits just above one workgroup's share went plain / equalised / plain ... every other build).
    Pure host logic; the Julia extension carries the same function (tests/test_host_cpu.py evaluates both)."""
    used, occupancy, eq, nohelp, ran_eq = word & 0xff, (word >> 8) & 0xff, (word >> 16) & 1, (word >> 17) & 1, (word >> 18) & 1
    sticky = 0
    if holdoff <= -2:
        if nohelp == 1:
            holdoff = EQ_HOLDOFF
        else:
            sticky = 1
            holdoff = holdoff + 1
            if holdoff == -1:
                holdoff = 0
    elif holdoff == -1:
        if used > 0:
            sticky = 1
            holdoff = -2 - EQ_STICKY
        else:
            holdoff = 0
    elif nohelp == 1 and holdoff == 0:
        # (bit 17: an equalised build found most of its records in crowded cells all the same — runs of equal keys — so the chain
        # stays with the plain grid for EQ_HOLDOFF rebuilds before it tries again)
        holdoff = EQ_HOLDOFF
    elif holdoff > 0:
        holdoff = holdoff - 1
    equalize = 1 if (used > 0 or eq == 1 or sticky == 1) and EQUALIZE and holdoff <= 0 else 0
    probation = 1 if ran_eq == 1 and equalize == 0 and holdoff == 0 and EQUALIZE else 0
    if probation == 1:
        holdoff = -1
    threshold = EQ_SPARE_OCCUPANCY if eq == 1 and equalize == 1 else SPARE_OCCUPANCY
    spare = 1 if used > 0 or occupancy >= threshold or n >= SPARE_ALWAYS_FROM or probation == 1 else 0
    return min(used + spare, abi.MAX_SORT_LEVELS), equalize, holdoff


_host_words_singleton = None


def _host_words():
    global _host_words_singleton
    if _host_words_singleton is None:
        _host_words_singleton = _HostWords()
    return _host_words_singleton


# ---------------------------------------------------------------------------------------------
# BVH
# ---------------------------------------------------------------------------------------------
class BVH:
    """struct BVH + constructor (build.jl:155-271).

        bvh = BVH(bounding_volumes, BBox(torch.float32), built_level=1, cache=None, options=BVHOptions())

    bounding_volumes: (n,4)/(n,6) CUDA float tensor, or a BoundingVolumes (kept indices, sorted IN PLACE,
    build.jl:128-153).  built_level: int level or float fraction (build.jl:309-325).  cache: a previous BVH
    whose nodes / skips / scratch buffers are reused (build.jl:232-238, 257-263).
    Fields: built_level, tree, skips, nodes, leaves (+ extrema: the expanded Morton bounds, device tensor).
    """

    def __init__(self, bounding_volumes, node_type=None, built_level=1, cache: Optional["BVH"] = None, options=None,
                 _out_of_place=False):
        torch = _require_gpu()
        # Time-stepping fast path: the same call as the one that built `cache` (raw volumes of the same shape and type,
        # default options, same node type and built level) repeats nothing but the build itself — every buffer, the tree
        # shape and the filled-in build descriptor are the cache's.  This is the host's critical path of a loop that reads
        # the contact count every step: the GPU idles from the moment the host sees the count until this call's first
        # launch arrives.
        fast = getattr(cache, "_fast", None) if cache is not None else None
        if (fast is not None and options is None and not _out_of_place and not isinstance(bounding_volumes, BoundingVolumes)
                and fast[0] == (bounding_volumes.shape, bounding_volumes.dtype, bounding_volumes.device, node_type, built_level)
                and bounding_volumes.is_cuda and bounding_volumes.is_contiguous() and cache._skew.valid()
                and bounding_volumes.data_ptr() != cache.leaves.buf.data_ptr()):
            d = fast[1]
            self.tree, self.built_level, self.types = cache.tree, cache.built_level, cache.types
            self.skips, self.nodes, self._scratch = cache.skips, cache.nodes, cache._scratch
            self.leaves, self.extrema, self._skew, self._fast = cache.leaves, cache.extrema, cache._skew, fast
            self._skew.apply(d)
            lib.call("ibvh_build", C.byref(d), _ptr(bounding_volumes), _ptr(self.leaves.buf), _ptr(self.nodes), _ptr(self.skips),
                     _ptr(self.extrema), _ptr(self._scratch), self._scratch.numel(), _stream())
            return
        # The reference's literal time-stepping loop — `bvh = BVH(bvh.leaves, N; cache=bvh)` after the leaves were moved IN
        # PLACE (build.jl:109-126, README.md:84-95): pre-wrapped records, user indices kept, the record array is input and
        # output.  Same idea: nothing but the build call itself is repeated.
        if (fast is not None and options is None and not _out_of_place and isinstance(bounding_volumes, BoundingVolumes)
                and bounding_volumes is cache.leaves and fast[0] == ("in place", bounding_volumes.buf.data_ptr(), len(bounding_volumes),
                                                                     node_type, built_level) and cache._skew.valid()):
            d = fast[1]
            self.tree, self.built_level, self.types = cache.tree, cache.built_level, cache.types
            self.skips, self.nodes, self._scratch = cache.skips, cache.nodes, cache._scratch
            self.leaves, self.extrema, self._skew, self._fast = cache.leaves, cache.extrema, cache._skew, fast
            self._skew.apply(d)
            lib.call("ibvh_build", C.byref(d), C.c_void_p(0), _ptr(self.leaves.buf), _ptr(self.nodes), _ptr(self.skips),
                     _ptr(self.extrema), _ptr(self._scratch), self._scratch.numel(), _stream())
            return
        fast_key = None
        if options is None and not _out_of_place and not isinstance(bounding_volumes, BoundingVolumes):
            fast_key = (bounding_volumes.shape, bounding_volumes.dtype, bounding_volumes.device, node_type, built_level)
        elif options is None and not _out_of_place:
            fast_key = ("in place", bounding_volumes.buf.data_ptr(), len(bounding_volumes), node_type, built_level)
        options = options or BVHOptions()
        node_type = node_type or BBox(torch.float32)  # default BBox{Float32} (build.jl:200)
        wrapped = isinstance(bounding_volumes, BoundingVolumes)
        if wrapped:
            src = bounding_volumes
            # check_bounding_volume_types (build.jl:355-361)
            if src.types.index_type != options.index_code:
                raise ValueError("BoundingVolume index type does not match BVHOptions index_exemplar type")
            if src.types.morton_type != options.morton_code:
                raise ValueError("BoundingVolume morton type does not match BVHOptions morton type")
            kind, flt, n = src.types.leaf_kind, src.types.leaf_float, len(src)
        else:
            kind, flt = _volume_kind(bounding_volumes)
            n = bounding_volumes.shape[0]
        types = abi.make_types(kind, flt, node_type.kind, node_type.flt, options.index_code, options.morton_code)
        if not abi.combo_supported(types):
            raise ValueError(f"no conversion from {'BSphere' if kind == abi.BSPHERE else 'BBox'} leaves to {node_type!r}")
        self.tree = _memo(("tree", int(n)), lambda: ImplicitTree(n))  # DomainError for n < 1
        tree = self.tree
        # compute_build_level (build.jl:309-325)
        if isinstance(built_level, (int, np.integer)):
            if not 1 <= built_level <= tree.levels:
                raise ValueError("1 <= built_level <= tree.levels must hold")
            built_ilevel = int(built_level)
        elif isinstance(built_level, float):
            out = C.c_int64()
            lib.call("ibvh_compute_build_level", C.byref(tree._t), float(built_level), C.byref(out))
            built_ilevel = out.value
        else:
            raise TypeError("built_level (the level to build BVH up to) must be an Integer or AbstractFloat")
        self.built_level = built_ilevel
        self.types = types
        idt, ndt = _torch_index(types.index_type), _torch_float(types.node_float)
        nw = abi.volume_width(types.node_kind)
        num_nodes = tree.real_nodes - tree.real_leaves
        # buffers, honouring cache= (type checks as build.jl:235,260)
        if cache is None:
            self.skips = torch.empty(tree.levels, dtype=idt, device="cuda")
            self.nodes = torch.empty((num_nodes, nw), dtype=ndt, device="cuda")
            self._scratch = None
        else:
            if cache.skips.dtype != idt:
                raise ValueError("eltype(cache.skips) === I must hold")
            if cache.nodes.dtype != ndt or cache.nodes.shape[1] != nw:
                raise ValueError("eltype(cache.nodes) === N must hold")
            self.skips = cache.skips if cache.skips.numel() == tree.levels else torch.empty(tree.levels, dtype=idt, device="cuda")
            self.nodes = cache.nodes if cache.nodes.shape[0] == num_nodes else torch.empty((num_nodes, nw), dtype=ndt, device="cuda")
            self._scratch = cache._scratch
        def scratch_need():
            need = C.c_size_t()
            lib.call("ibvh_build_scratch_bytes", C.byref(types), n, C.byref(need))
            return need.value
        need = _memo(("build_scratch", types.key(), int(n)), scratch_need)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device="cuda")
        if wrapped and _out_of_place:
            # source records stay untouched, the sorted records go to a (reused) buffer of their own
            nbytes = n * int(_layout_of(types).leaf_bytes)
            reuse = cache is not None and cache.leaves.buf.numel() == nbytes and cache.leaves.buf.data_ptr() != bounding_volumes.buf.data_ptr()
            self.leaves = BoundingVolumes(types, n, cache.leaves.buf if reuse else torch.empty(nbytes, dtype=torch.uint8, device="cuda"))
            vol_ptr = _ptr(bounding_volumes.buf)
        elif wrapped:
            self.leaves = bounding_volumes
            self.leaves.types = types  # node type is not part of the record layout
            vol_ptr = C.c_void_p(0)
        else:
            # raw volumes: the wrapped, sorted leaves are this BVH's own array — the cached one when it has the right size
            # and type (the reference allocates a new one every time, build.jl:345-349; with cache= a time step then
            # allocates nothing at all: nodes, skips, leaves, extrema and scratch are all reused)
            nbytes = n * int(_layout_of(types).leaf_bytes)
            reuse = (cache is not None and cache.leaves.buf.numel() == nbytes and cache.leaves.types.key() == types.key()
                     and cache.leaves.buf.data_ptr() != bounding_volumes.data_ptr())
            self.leaves = BoundingVolumes(types, n, cache.leaves.buf if reuse else torch.empty(nbytes, dtype=torch.uint8, device="cuda"))
            bounding_volumes = bounding_volumes.contiguous()
            vol_ptr = _ptr(bounding_volumes)
        ext_dt = _torch_float(flt)
        self.extrema = cache.extrema if (cache is not None and cache.extrema is not None and cache.extrema.dtype == ext_dt) \
            else torch.empty(6, dtype=ext_dt, device="cuda")
        # skewed inputs: a cold build launches COLD_SORT_LEVELS extra partition levels of the sort; a build that reuses
        # `cache=` launches as many as the previous build of the chain would have used, plus a spare one (none at all
        # after a uniform cloud).  The GPU leaves that number in a pinned host word (mapped into the device's address
        # space), which is read here WITHOUT synchronising: the latest value that has arrived is good enough for a
        # hint (include/ibvh.h, ibvh_build_desc.sort_levels / skew_flag).
        if cache is not None and getattr(cache, "_skew", None) is not None and cache._skew.valid():
            self._skew = cache._skew  # the chain's hint word
        else:
            self._skew = _HintWord(COLD_SORT_LEVELS)
        d = abi.BuildDesc()
        d.types = types
        d.n = n
        d.built_level = built_ilevel
        d.already_wrapped = 1 if wrapped else 0
        alg = options.morton
        d.compute_extrema = 1 if alg.compute_extrema else 0
        if not alg.compute_extrema:
            d.mins[:] = [float(v) for v in alg.mins]
            d.maxs[:] = [float(v) for v in alg.maxs]
        # A SPARE extra level (four launches that find nothing to do: ~15 us per step at 1e6 leaves) whenever the chain is
        # anywhere near needing one: the previous build used extra levels, or its fullest cell was beyond SPARE_OCCUPANCY
        # of what a finish workgroup sorts — a cloud that contracts or clusters over many steps reaches that long before
        # a cell overflows.  Only an input that changes ABRUPTLY from comfortably uniform to clustered meets no extra
        # level: its crowded cells take the one-workgroup slow path on that one step (correct; 17 ms at 1e6 leaves,
        # tools/attic/dbg_spike.py) and the hint it leaves fixes the next.  That path's cost grows with the cell (~0.1 s at 1e7
        # leaves) while the idle level's share of a step shrinks (1 % at 1e7), so builds of SPARE_ALWAYS_FROM leaves and
        # more always launch the spare level.  SPARE_OCCUPANCY = 0 restores "always" at every size.
        # EQUALISED cells (include/ibvh.h, sort_equalize) once the chain's input has shown that it does not fill the grid: the
        # previous build needed extra levels, or — itself built with equalised cells — reported that the plain grid would still
        # have had a crowded cell (bit 16 of the hint).  A cloud that fills its box never pays the two extra launches.
        if cache is None:
            d.sort_levels, d.sort_equalize = COLD_SORT_LEVELS, 0
        else:
            self._skew.apply(d)
        d.skew_flag = self._skew.ptr()
        lib.call("ibvh_build", C.byref(d), vol_ptr, _ptr(self.leaves.buf), _ptr(self.nodes), _ptr(self.skips),
                 _ptr(self.extrema), _ptr(self._scratch), self._scratch.numel(), _stream())
        self._fast = (fast_key, d) if fast_key is not None else None

    @classmethod
    def from_buffers(cls, types, n, leaves_buf, nodes, built_level=1):
        """Wrap existing device buffers (sorted BoundingVolume records + node array, e.g. received from a peer
        GPU) as a BVH without building anything; skips are recomputed from the tree shape."""
        torch = _require_gpu()
        self = cls.__new__(cls)
        self.types = types
        self.tree = ImplicitTree(n)
        self.built_level = int(built_level)
        self.leaves = BoundingVolumes(types, n, leaves_buf)
        self.nodes = nodes
        sk = (C.c_int64 * self.tree.levels)()
        lib.call("ibvh_compute_skips", C.byref(self.tree._t), sk)
        self.skips = torch.tensor(list(sk), dtype=_torch_index(types.index_type), device="cuda")
        self.extrema = None
        self._scratch = None
        self._skew = None
        return self

    def struct(self):
        b = abi.Bvh()
        b.types = self.types
        b.tree = self.tree._t
        b.built_level = self.built_level
        b.leaves = self.leaves.buf.data_ptr()
        b.nodes = self.nodes.data_ptr() if self.nodes.numel() else 0
        b.skips = self.skips.data_ptr()
        return b

    def __repr__(self):
        return (f"BVH\n  built_level: {self.built_level}\n  tree:        {self.tree}\n  skips:       {tuple(self.skips.shape)}\n"
                f"  nodes:       {tuple(self.nodes.shape)}\n  leaves:      ({len(self.leaves)},)\n")


def default_start_level(bvh, alg=None):
    """lvt/leaf_vs_tree.jl:4-6 and bfs/breadth_first.jl:4-6."""
    if alg is None or isinstance(alg, LVTTraversal):
        return max(1, bvh.built_level)
    if isinstance(alg, BFSTraversal):
        return max(bvh.tree.levels // 2, bvh.built_level)
    raise ValueError(f"default_start_level not implemented for: {alg}")


# ---------------------------------------------------------------------------------------------
# traversal results
# ---------------------------------------------------------------------------------------------
class BVHTraversal:
    """traverse.jl:54-107.  cache1 holds the contacts ((m, 2) index tensor); cache2 is the other buffer
    (LVT: per-work-item inclusive counts; BFS: the second pair queue).  `.contacts` = cache1[:num_contacts].

    LVT traversals that were given a large-enough `cache` are enqueued without any host synchronisation
    (ibvh_traverse_*_lvt_enqueue): `num_contacts` is then read from the device the first time it is asked for
    (that read is the reference's `@allowscalar`, moved to where the value is needed).  If the contact buffer
    taken from the cache turns out too small, the writing pass runs then, into a new buffer — which needs the
    traversal's counts and scratch to be intact, i.e. not handed to a later call as `cache=` in between."""

    def __init__(self, start_level1, start_level2, num_checks, num_contacts, cache1, cache2, _scratch=None, _pending=None):
        self.start_level1 = start_level1
        self.start_level2 = start_level2
        self.num_checks = num_checks
        self._num_contacts = num_contacts
        self._cache1 = cache1
        self.cache2 = cache2
        self._scratch = _scratch
        self._pending = _pending  # (total: _PendingTotal, capacity, finish(total) -> contacts tensor)
        self._donated = False

    def _resolve(self):
        if self._pending is None:
            return
        total_t, capacity, finish = self._pending
        self._pending = None
        total = int(total_t.item())  # the blocking read (the pinned word: published before the scan / writing pass finish)
        if hasattr(total_t, "order_current_stream"):
            total_t.order_current_stream()  # a consumer on another stream waits for the launch stream
        if self._cache1.dtype == _torch().int32 and total > 2**31 - 1:
            raise OverflowError("more than typemax(Int32) contacts")
        if total > capacity:
            if self._donated:
                raise RuntimeError("this traversal's buffers were reused as `cache=` before its contact count was read, and "
                                   "the cached contact buffer was too small for it: read .num_contacts before reusing the cache")
            self._cache1 = finish(total)
        self._num_contacts = total

    @property
    def num_contacts(self):
        self._resolve()
        return self._num_contacts

    @property
    def cache1(self):
        self._resolve()
        return self._cache1

    @property
    def contacts(self):
        return self.cache1[: self.num_contacts]

    def _capacity(self):
        """rows of the contact buffer, without forcing the count to be read"""
        return self._cache1.shape[0] if self._cache1 is not None else 0

    def _donate(self):
        """the buffers are about to be reused by another traversal"""
        self._donated = True
        return self._cache1


class LeafBatch:
    """What a `narrow` callable sees for one side of m candidate pairs, as tensors on the GPU: .volume (m, 4|6),
    .index (m,), .morton (m,) int64 — the fields of the reference's BoundingVolume (bounding_volumes.jl:55-59), batched."""

    def __init__(self, bvh, positions):
        rows = positions.long() - 1
        self.volume = bvh.leaves.volume[rows]
        self.index = bvh.leaves.index[rows]
        self.morton = bvh.leaves.morton_device[rows]

    def __len__(self):
        return int(self.index.shape[0])


def _narrow_code(narrow, rays=False):
    """-> (code for the C ABI, callable or None).  The reference takes any closure (traverse.jl:213, raytrace.jl:76) and
    only ever evaluates it as `iscontact(...) && narrow(...)` at leaf level, i.e. as a post-filter.  Menu entries run on
    the device; any other callable — VECTORISED: narrow(a: LeafBatch, b: LeafBatch) -> bool tensor, for rays
    narrow(bv: LeafBatch, points (m, 3), directions (m, 3)) — is applied by this mirror to the candidate list the device
    returns as leaf positions (IBVH_OUTPUT_POSITIONS, include/ibvh.h)."""
    if narrow is None:
        return abi.NARROW_NONE, None
    menu = (abi.NARROW_NONE, abi.NARROW_RAY_ORIGIN_OUTSIDE) if rays else (abi.NARROW_NONE, abi.NARROW_MORTON_LT, abi.NARROW_INDEX_LT)
    if isinstance(narrow, (int, np.integer)) and not isinstance(narrow, bool):
        if int(narrow) in menu:
            return int(narrow), None
        raise ValueError(f"narrow = {narrow} is not on the {'ray' if rays else 'pair'} menu of include/ibvh.h")
    if callable(narrow):
        return abi.OUTPUT_POSITIONS, narrow
    raise TypeError("narrow must be None, a menu constant (NARROW_*) or a vectorised callable")


def _post_filter(trav, narrow, bvh1, bvh2=None, rays=None):
    """Apply a callable `narrow` to a traversal whose contact list holds leaf POSITIONS; returns the traversal with the
    surviving contacts as user indices, order preserved (so LVT results keep the reference's order)."""
    torch = _torch()
    pos = trav.contacts
    if pos.shape[0] == 0:
        keep = torch.zeros(0, dtype=torch.bool, device=pos.device)
    if rays is not None:
        p, d = rays  # (N, 3) row-major
        a = LeafBatch(bvh1, pos[:, 0])
        r = pos[:, 1].long() - 1
        keep = narrow(a, p[r], d[r]) if pos.shape[0] else keep
        out = torch.stack([a.index, pos[:, 1]], dim=1)
    else:
        a = LeafBatch(bvh1, pos[:, 0])
        b = LeafBatch(bvh2 if bvh2 is not None else bvh1, pos[:, 1])
        keep = narrow(a, b) if pos.shape[0] else keep
        if bvh2 is None:  # (min, max) of the user indices, traverse_single.jl:176-180
            out = torch.stack([torch.minimum(a.index, b.index), torch.maximum(a.index, b.index)], dim=1)
        else:
            out = torch.stack([a.index, b.index], dim=1)
    keep = torch.as_tensor(keep, device=pos.device).to(torch.bool).reshape(-1)
    if keep.shape[0] != pos.shape[0]:
        raise ValueError("narrow must return one bool per candidate pair")
    contacts = out[keep].to(pos.dtype).contiguous()
    return BVHTraversal(trav.start_level1, trav.start_level2, trav.num_checks, int(contacts.shape[0]), contacts, trav.cache2)


def _cache_tensor(cache_t, need_rows, cols, dtype, what):
    """Reuse a cache buffer if it is large enough (resize! semantics), checking its eltype (@argcheck)."""
    torch = _torch()
    shape = (need_rows, cols) if cols else (need_rows,)
    if cache_t is None:
        return torch.empty(shape, dtype=dtype, device="cuda")
    if cache_t.dtype != dtype or (cols and (cache_t.dim() != 2 or cache_t.shape[1] != cols)) or (not cols and cache_t.dim() != 1):
        raise ValueError(f"eltype(cache.{what}) does not match the traversal's index type")
    if cache_t.shape[0] < need_rows:
        return torch.empty(shape, dtype=dtype, device="cuda")
    return cache_t


# Equalised builds re-fit their cells to the input every time, so a chain's occupancy does not creep up on them: a cell overflows only
# by sampling chance (32 samples per cell at 1e6 leaves: never, in practice) or through runs of equal keys (then `used` > 0 and the
# levels are launched anyway), and a mildly overflowing cell without a level costs one slower finish workgroup (~15 us), what the
# idle spare level (plan + four launches) costs EVERY step.  So they launch a spare level only when the fullest cell was about full.
EQ_SPARE_OCCUPANCY = 120
EQ_HOLDOFF = 32  # rebuilds a chain stays with the plain grid after an equalised build reported that it did not help (hint bit 17)
EQ_STICKY = 32   # rebuilds a chain stays with equalised cells after a return to the plain grid turned out crowded (sort_hint_rule)
EQUALIZE = True  # rebuilds of a chain whose input is skewed ask for equalised cells (False: the regular grid + extra levels, always)
COLD_SORT_LEVELS = 2  # extra partition levels a build without cache= launches (include/ibvh.h, sort_levels)
SPARE_OCCUPANCY = 96  # (of 128) fullest coarse cell from which a cached build launches a spare extra level (BVH.__init__)
SPARE_ALWAYS_FROM = 1 << 24  # leaves from which a cached build always launches it (the slow path it avoids grows with the input,
                             # the idle level's share of a step shrinks: 0.7 % at 1.7e7 leaves)
LVT_CACHE_SLOTS = 8  # contacts per work item kept from the counting pass (include/ibvh.h)
RAY_CACHE_SLOTS = 32  # hits per ray kept from the counting pass


def _speculative_buffer(cache, idt):
    """The cached contact buffer, if the traversal may be enqueued against it without knowing the count."""
    if cache is None or not isinstance(cache, BVHTraversal):
        return None
    buf = cache._donate()
    if buf is None or buf.dim() != 2 or buf.shape[1] != 2 or buf.shape[0] == 0:
        return None
    if buf.dtype != idt:
        raise ValueError("eltype(cache.cache1) does not match the traversal's index type")
    return buf


class _LvtScratch:
    """The LVT scratch buffer of a chain of traversals that hand each other their buffers through `cache=`, plus a ring
    of device-side totals.

    A traversal that was enqueued without a host read (`*_enqueue`) fetches its total contact count later, possibly
    after later traversals have reused — and rewritten — the scratch buffer (header, tile sums and contact cache
    alike).  So the totals do NOT live in the scratch: every call hands the library its own int64 word of a separate
    64-entry ring (`total_dev` of include/ibvh.h), which nothing else writes until the ring wraps, 64 calls later; a
    total that was never read by then raises instead of returning another call's number.  The host normally gets
    the total from the pinned mirror word the scan kernel fills (`total_host`), without touching the stream."""
    SLOTS = 64

    def __init__(self, nbytes):
        torch = _torch()
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        self.totals = torch.zeros(self.SLOTS, dtype=torch.int64, device="cuda")
        self.calls = 0

    def capacity(self):
        return self.buf.numel()

    def data_ptr(self):
        return self.buf.data_ptr()

    def numel(self):
        return self.buf.numel()

    def next_total(self):
        """(device pointer of this call's total word, pinned host pointer of its mirror, handle to read it later)"""
        slot = self.calls % self.SLOTS
        self.calls += 1
        hw = _host_words()
        hslot, hgen = hw.take(_HostWords.PENDING)
        return C.c_void_p(self.totals.data_ptr() + 8 * slot), C.c_void_p(hw.ptr(hslot)), _PendingTotal(self, slot, self.calls, hslot, hgen)


class _PendingTotal:
    SPIN = 2_000_000  # polls of the pinned word before falling back to the event + device read (seconds of spinning)

    def __init__(self, owner, slot, calls, hslot, hgen):
        self.owner, self.slot, self.calls = owner, slot, calls
        self.hslot, self.hgen = hslot, hgen
        self.event = None

    def launched(self):
        """Remember the stream the traversal was enqueued on (the raw handle: no event is recorded on the hot path).  The
        pinned word is published BEFORE the scan and the guarded writing pass have finished, so a consumer on ANOTHER
        stream must be ordered behind this one explicitly: order_current_stream()."""
        self.stream = _stream().value or 0
        return self

    def order_current_stream(self):
        """Make torch's current stream wait for everything enqueued on the launch stream so far (a no-op when they are the
        same stream: in-order).  Called by BVHTraversal._resolve before the writing pass is re-launched or the contacts
        are handed out."""
        stream = getattr(self, "stream", None)
        if stream is None:
            return
        cur = _stream().value or 0
        if cur == stream:
            return
        torch = _torch()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream())
        torch.cuda.current_stream().wait_event(ev)

    def item(self):
        if self.owner.calls - self.calls >= self.owner.SLOTS:
            raise RuntimeError("this traversal's contact count was never read and its slot in the totals ring has been "
                               f"recycled by {self.owner.SLOTS} later traversals on the same cache")
        # The scan kernel stores the total into the pinned word as well (total_host): poll it — no stream
        # synchronisation, no device-to-host copy (the reference's blocking `@allowscalar` read costs both,
        # lvt/traverse_single.jl:60).
        hw = _host_words()
        if hw.generation[self.hslot] == self.hgen:
            words, k = hw.words, self.hslot
            for _ in range(self.SPIN):
                v = int(words[k])
                if v != _HostWords.PENDING:
                    return v
        # the word never arrived (not expected): wait for the launch stream, read the device copy
        stream = getattr(self, "stream", None)
        torch = _torch()
        if stream:
            torch.cuda.ExternalStream(stream).synchronize()
        else:
            torch.cuda.synchronize()
        return int(self.owner.totals[self.slot].item())


def _cache_slots(cache, n_items, default):
    """Contact-cache slots per work item: the default, or — when a previous traversal's contact buffer is being
    reused — about four times the contacts per item that buffer was sized for (rounded up to a power of two, at most
    64): a wave whose items together overflow their pooled slots has to walk twice (7 contacts per leaf: 0.50 ms with
    8 slots, 0.30 ms with 32 at 1e6 leaves).  Uses the buffer's size only, never forces the lazy count to be read."""
    if not isinstance(cache, BVHTraversal) or n_items <= 0:
        return default
    per_item = -(-cache._capacity() // int(n_items))  # ceil
    want = 1
    while want < 4 * per_item:
        want *= 2
    return max(default, min(64, want))


def _lvt_scratch(cache, types, n_items, slots=None, rays_bvh=None):
    k = _cache_slots(cache, n_items, LVT_CACHE_SLOTS if slots is None else slots)

    def scratch_need():
        need = C.c_size_t()
        if rays_bvh is not None:  # rays: room for the walker's shadow of the node levels as well (include/ibvh.h)
            s = rays_bvh.struct()
            lib.call("ibvh_rays_scratch_bytes", C.byref(s), int(n_items), k, C.byref(need))
        else:
            lib.call("ibvh_lvt_scratch_bytes", C.byref(types), int(n_items), k, C.byref(need))
        return need.value
    key = ("lvt_scratch", types.key(), int(n_items), k) if rays_bvh is None else \
        ("rays_scratch", types.key(), int(n_items), k, len(rays_bvh.leaves), rays_bvh.built_level)
    need = _memo(key, scratch_need)
    s = cache._scratch if cache is not None else None
    if not isinstance(s, _LvtScratch) or s.capacity() < need:
        s = _LvtScratch(need)
    return s


def _traverse_lvt_single(bvh, start_level, narrow, cache):
    torch = _require_gpu()
    idt = _torch_index(bvh.types.index_type)
    if not (bvh.built_level <= start_level <= bvh.tree.levels <= 32):
        raise ValueError("bvh.built_level <= start_level <= bvh.tree.levels <= 32 must hold")
    if bvh.tree.real_nodes <= 1:  # traverse_single.jl:17-21
        return BVHTraversal(start_level, 0, 0, 0, torch.empty((0, 2), dtype=idt, device="cuda"),
                            torch.empty(0, dtype=idt, device="cuda"))
    n = len(bvh.leaves)
    counts = _cache_tensor(cache.cache2 if cache else None, n, 0, idt, "cache2")
    scratch = _lvt_scratch(cache, bvh.types, n)
    sp, sn = _ptr(scratch), scratch.numel()
    s = bvh.struct()
    spec = _speculative_buffer(cache, idt)
    if spec is not None:
        tdev, thost, pending = scratch.next_total()
        lib.call("ibvh_traverse_lvt_enqueue", C.byref(s), start_level, narrow, _ptr(counts), _ptr(spec), spec.shape[0],
                 tdev, thost, sp, sn, _stream())
        pending.launched()

        def finish(total):
            contacts = torch.empty((total, 2), dtype=idt, device="cuda")
            lib.call("ibvh_traverse_lvt_write", C.byref(s), start_level, narrow, _ptr(counts), _ptr(contacts), sp,
                     sn, _stream())
            return contacts
        return BVHTraversal(start_level, 0, 0, None, spec, counts, scratch, _pending=(pending, spec.shape[0], finish))
    total = C.c_int64()
    lib.call("ibvh_traverse_lvt_count", C.byref(s), start_level, narrow, _ptr(counts), C.byref(total), sp,
             sn, _stream())
    contacts = _cache_tensor(cache.cache1 if cache else None, total.value, 2, idt, "cache1")
    if total.value:
        lib.call("ibvh_traverse_lvt_write", C.byref(s), start_level, narrow, _ptr(counts), _ptr(contacts), sp,
                 sn, _stream())
    return BVHTraversal(start_level, 0, 0, total.value, contacts, counts, scratch)


def _traverse_lvt_pair(bvh1, bvh2, sl1, sl2, narrow, cache):
    torch = _require_gpu()
    if bvh1.types.index_type != bvh2.types.index_type:
        raise ValueError("get_index_type(bvh2) === I must hold")  # traverse_pair.jl:50-52
    idt = _torch_index(bvh1.types.index_type)
    for b, sl in ((bvh1, sl1), (bvh2, sl2)):
        if not (b.built_level <= sl <= b.tree.levels <= 32):
            raise ValueError("bvh.built_level <= start_level <= bvh.tree.levels <= 32 must hold")
    n = max(len(bvh1.leaves), len(bvh2.leaves))
    counts = _cache_tensor(cache.cache2 if cache else None, n, 0, idt, "cache2")
    scratch = _lvt_scratch(cache, bvh1.types, n)
    sp, sn = _ptr(scratch), scratch.numel()
    s1, s2 = bvh1.struct(), bvh2.struct()
    spec = _speculative_buffer(cache, idt)
    if spec is not None:
        tdev, thost, pending = scratch.next_total()
        lib.call("ibvh_traverse_pair_lvt_enqueue", C.byref(s1), C.byref(s2), sl1, sl2, narrow, _ptr(counts), _ptr(spec),
                 spec.shape[0], tdev, thost, sp, sn, _stream())
        pending.launched()

        def finish(total):
            contacts = torch.empty((total, 2), dtype=idt, device="cuda")
            lib.call("ibvh_traverse_pair_lvt_write", C.byref(s1), C.byref(s2), sl1, sl2, narrow, _ptr(counts), _ptr(contacts),
                     sp, sn, _stream())
            return contacts
        return BVHTraversal(sl1, sl2, 0, None, spec, counts, scratch, _pending=(pending, spec.shape[0], finish))
    total = C.c_int64()
    lib.call("ibvh_traverse_pair_lvt_count", C.byref(s1), C.byref(s2), sl1, sl2, narrow, _ptr(counts), C.byref(total),
             sp, sn, _stream())
    contacts = _cache_tensor(cache.cache1 if cache else None, total.value, 2, idt, "cache1")
    if total.value:
        lib.call("ibvh_traverse_pair_lvt_write", C.byref(s1), C.byref(s2), sl1, sl2, narrow, _ptr(counts), _ptr(contacts),
                 sp, sn, _stream())
    return BVHTraversal(sl1, sl2, 0, total.value, contacts, counts, scratch)


BFS_INITIAL_FACTOR = 4  # queues start at 4x the initial pair count (bfs/traverse_single.jl:73)
_last_bfs_counters = None
BFS_GROWTH = 4          # and grow at least 4x when a level does not fit (deeper levels need more still: 2x cost twice the resumes)


def _bfs_run(entry, types, initial_capacity, cache, levels_hint, *args):
    """Shared BFS driver.  Queues start at 4x the initial pair count (the reference's sizing, bfs/traverse_single.jl:73)
    or at the size of the cached buffers; when the library reports IBVH_ERR_CAPACITY both queues are grown (the pairs of
    the level that overflowed are kept) and the SAME traversal continues from that level — the reference's resize!
    between two levels (bfs/traverse_single.jl:38-53) — instead of starting over."""
    torch = _require_gpu()
    idt = _torch_index(types.index_type)
    cap = BFS_INITIAL_FACTOR * max(int(initial_capacity), 1)
    q1 = cache.cache1 if cache else None
    q2 = cache.cache2 if cache else None
    need = C.c_size_t()
    lib.call("ibvh_bfs_counters_bytes", int(levels_hint), C.byref(need))
    counters = torch.zeros(need.value, dtype=torch.uint8, device="cuda")
    q1 = _cache_tensor(q1, cap, 2, idt, "cache1")
    q2 = _cache_tensor(q2, cap, 2, idt, "cache2")
    res = abi.BfsResult()
    while True:
        cap = min(q1.shape[0], q2.shape[0])
        st = getattr(lib.load(), entry)(*args, _ptr(q1), _ptr(q2), cap, _ptr(counters), C.byref(res), _stream())
        if st == abi.ERR_CAPACITY:
            new_cap = max(int(res.required_capacity), BFS_GROWTH * cap)
            keep = int(res.resume_num) if res.resume_step > 0 else 0
            grown = []
            for k, q in enumerate((q1, q2), start=1):
                g = torch.empty((new_cap, 2), dtype=idt, device="cuda")
                if keep and k == res.contacts_in:
                    g[:keep].copy_(q[:keep])  # the pairs the overflowed level still has to expand
                grown.append(g)
            q1, q2 = grown
            continue
        abi.check(st, entry)
        if res.contacts_in == 2:
            q1, q2 = q2, q1
        global _last_bfs_counters
        _last_bfs_counters = (counters, int(levels_hint))  # (tools/attic/dbg_bfs_levels.py reads the per-step counts)
        return res, q1, q2


def _traverse_bfs_single(bvh, start_level, narrow, cache):
    torch = _require_gpu()
    idt = _torch_index(bvh.types.index_type)
    if not (bvh.tree.levels >= start_level >= bvh.built_level):
        raise ValueError("bvh.tree.levels >= start_level >= bvh.built_level must hold")
    if bvh.tree.real_nodes <= 1:
        e = torch.empty((0, 2), dtype=idt, device="cuda")
        return BVHTraversal(start_level, 0, 0, 0, e, e.clone())
    s = bvh.struct()
    cap = C.c_int64()
    lib.call("ibvh_bfs_initial_capacity", C.byref(s), start_level, C.byref(cap))
    res, q1, q2 = _bfs_run("ibvh_traverse_bfs", bvh.types, cap.value, cache, bvh.tree.levels, C.byref(s), start_level, narrow)
    return BVHTraversal(start_level, 0, res.num_checks, res.num_contacts, q1, q2)


def _traverse_bfs_pair(bvh1, bvh2, sl1, sl2, narrow, cache):
    _require_gpu()
    for b, sl in ((bvh1, sl1), (bvh2, sl2)):
        if not (b.tree.levels >= sl >= b.built_level):
            raise ValueError("bvh.tree.levels >= start_level >= bvh.built_level must hold")
    s1, s2 = bvh1.struct(), bvh2.struct()
    cap = C.c_int64()
    lib.call("ibvh_bfs_pair_initial_capacity", C.byref(s1), C.byref(s2), sl1, sl2, C.byref(cap))
    res, q1, q2 = _bfs_run("ibvh_traverse_pair_bfs", bvh1.types, cap.value, cache, bvh1.tree.levels + bvh2.tree.levels,
                           C.byref(s1), C.byref(s2), sl1, sl2, narrow)
    return BVHTraversal(sl1, sl2, res.num_checks, res.num_contacts, q1, q2)


def traverse(bvh, *args, start_level=None, start_level1=None, start_level2=None, narrow=None, cache=None, options=None):
    """traverse(bvh, alg=LVTTraversal(); start_level, narrow, cache, options)
       traverse(bvh1, bvh2, alg=LVTTraversal(); start_level1, start_level2, narrow, cache, options)
    (traverse/traverse.jl:210-233).  Returns a BVHTraversal; `.contacts` is an (m, 2) tensor of 1-based index
    pairs: (min, max) of the user indices for one BVH, (index in bvh1, index in bvh2) for two."""
    bvh2, alg = None, None
    for a in args:
        if isinstance(a, BVH):
            bvh2 = a
        elif isinstance(a, (LVTTraversal, BFSTraversal)):
            alg = a
        else:
            raise ValueError(f"Traversal algorithm not implemented: {a}")  # traverse.jl:217
    alg = alg or LVTTraversal()
    code, fn = _narrow_code(narrow)
    if bvh2 is None:
        sl = default_start_level(bvh, alg) if start_level is None else int(start_level)
        if isinstance(alg, LVTTraversal):
            t = _traverse_lvt_single(bvh, sl, code, cache)
        else:
            t = _traverse_bfs_single(bvh, sl, code, cache)
        return _post_filter(t, fn, bvh) if fn else t
    sl1 = default_start_level(bvh, alg) if start_level1 is None else int(start_level1)
    sl2 = default_start_level(bvh2, alg) if start_level2 is None else int(start_level2)
    if isinstance(alg, LVTTraversal):
        t = _traverse_lvt_pair(bvh, bvh2, sl1, sl2, code, cache)
    else:
        t = _traverse_bfs_pair(bvh, bvh2, sl1, sl2, code, cache)
    return _post_filter(t, fn, bvh, bvh2) if fn else t


def traverse_rays(bvh, points, directions, alg=None, start_level=1, narrow=None, cache=None, options=None):
    """traverse_rays(bvh, points, directions, alg=LVTTraversal(); start_level=1, narrow, cache, options)
    (raytrace/raytrace.jl:71-80).  points, directions: (3, N) tensors as in the reference (column i = ray i);
    contacts are (leaf.index, iray).  Rays are converted to the leaf float type (raytrace/lvt:116-125)."""
    torch = _require_gpu()
    alg = alg or LVTTraversal()
    code, fn = _narrow_code(narrow, rays=True)
    if not (points.dim() == 2 and directions.dim() == 2 and points.shape[0] == 3 and directions.shape[0] == 3):
        raise ValueError("size(points, 1) == size(directions, 1) == 3 must hold")
    if points.shape[1] != directions.shape[1]:
        raise ValueError("size(points, 2) == size(directions, 2) must hold")
    lvt = isinstance(alg, LVTTraversal)
    if lvt and not (bvh.built_level <= start_level <= bvh.tree.levels <= 32):
        raise ValueError("bvh.built_level <= start_level <= bvh.tree.levels <= 32 must hold")
    if not lvt and not (bvh.tree.levels >= start_level >= bvh.built_level):
        raise ValueError("bvh.tree.levels >= start_level >= bvh.built_level must hold")
    idt = _torch_index(bvh.types.index_type)
    nr = points.shape[1]
    if nr == 0:
        e = torch.empty((0, 2), dtype=idt, device="cuda")
        return BVHTraversal(start_level, 0, 0, 0, e, e.clone())
    ft = _torch_float(bvh.types.leaf_float)
    p = points.to(device="cuda", dtype=ft).t().contiguous()  # (N, 3) row-major == (3, N) column-major
    d = directions.to(device="cuda", dtype=ft).t().contiguous()
    s = bvh.struct()
    if lvt:
        counts = _cache_tensor(cache.cache2 if cache else None, nr, 0, idt, "cache2")
        scratch = _lvt_scratch(cache, bvh.types, nr, slots=RAY_CACHE_SLOTS, rays_bvh=bvh)
        sp, sn = _ptr(scratch), scratch.numel()
        spec = _speculative_buffer(cache, idt)
        if spec is not None:
            tdev, thost, pending = scratch.next_total()
            lib.call("ibvh_traverse_rays_lvt_enqueue", C.byref(s), _ptr(p), _ptr(d), nr, start_level, code, _ptr(counts), _ptr(spec),
                     spec.shape[0], tdev, thost, sp, sn, _stream())
            pending.launched()

            def finish(total):
                contacts = torch.empty((total, 2), dtype=idt, device="cuda")
                lib.call("ibvh_traverse_rays_lvt_write", C.byref(s), _ptr(p), _ptr(d), nr, start_level, code, _ptr(counts),
                         _ptr(contacts), sp, sn, _stream())
                return contacts
            t = BVHTraversal(start_level, 0, 0, None, spec, counts, scratch, _pending=(pending, spec.shape[0], finish))
            return _post_filter(t, fn, bvh, rays=(p, d)) if fn else t
        total = C.c_int64()
        lib.call("ibvh_traverse_rays_lvt_count", C.byref(s), _ptr(p), _ptr(d), nr, start_level, code, _ptr(counts),
                 C.byref(total), sp, sn, _stream())
        contacts = _cache_tensor(cache.cache1 if cache else None, total.value, 2, idt, "cache1")
        if total.value:
            lib.call("ibvh_traverse_rays_lvt_write", C.byref(s), _ptr(p), _ptr(d), nr, start_level, code, _ptr(counts),
                     _ptr(contacts), sp, sn, _stream())
        t = BVHTraversal(start_level, 0, 0, total.value, contacts, counts, scratch)
        return _post_filter(t, fn, bvh, rays=(p, d)) if fn else t
    cap = C.c_int64()
    lib.call("ibvh_bfs_rays_initial_capacity", C.byref(s), nr, start_level, C.byref(cap))
    res, q1, q2 = _bfs_run("ibvh_traverse_rays_bfs", bvh.types, cap.value, cache, bvh.tree.levels, C.byref(s), _ptr(p),
                           _ptr(d), nr, start_level, code)
    t = BVHTraversal(start_level, 0, res.num_checks, res.num_contacts, q1, q2)
    return _post_filter(t, fn, bvh, rays=(p, d)) if fn else t


def lvt_work_counters(bvh, bvh2=None, points=None, directions=None):
    """Measurement helper (ibvh_lvt_work_counters, include/ibvh.h): the work of ONE counting pass of the leaf-vs-tree walk
    the ordinary call would take — traverse(bvh), traverse(bvh, bvh2) or traverse_rays(bvh, points, directions) — as a dict
    {node_tests, leaf_tests, node_fetches, leaf_fetches, touched_bytes}.  Bench types only (the reference
    counts such work for BFS only, BVHTraversal.num_checks)."""
    torch = _require_gpu()
    idt = _torch_index(bvh.types.index_type)
    work = torch.zeros(4, dtype=torch.int64, device="cuda")
    s = bvh.struct()
    if points is not None:
        ft = _torch_float(bvh.types.leaf_float)
        p = points.to(device="cuda", dtype=ft).t().contiguous()
        d = directions.to(device="cuda", dtype=ft).t().contiguous()
        counts = torch.empty(p.shape[0], dtype=idt, device="cuda")
        lib.call("ibvh_lvt_work_counters", C.byref(s), None, _ptr(p), _ptr(d), p.shape[0], _ptr(counts), _ptr(work), _stream())
    elif bvh2 is not None:
        s2 = bvh2.struct()
        counts = torch.empty(max(len(bvh.leaves), len(bvh2.leaves)), dtype=idt, device="cuda")
        lib.call("ibvh_lvt_work_counters", C.byref(s), C.byref(s2), None, None, 0, _ptr(counts), _ptr(work), _stream())
    else:
        counts = torch.empty(len(bvh.leaves), dtype=idt, device="cuda")
        lib.call("ibvh_lvt_work_counters", C.byref(s), None, None, None, 0, _ptr(counts), _ptr(work), _stream())
    w = [int(x) for x in work.cpu().tolist()]
    lay = bvh.leaves.layout
    return {"node_tests": w[0], "leaf_tests": w[1], "node_fetches": w[2], "leaf_fetches": w[3],
            "touched_bytes": w[2] * int(lay.node_bytes) + w[3] * int(lay.leaf_bytes), "contacts_counted": int(counts.sum().item())}


# ---------------------------------------------------------------------------------------------
# input preparation
# ---------------------------------------------------------------------------------------------
def bounding_volumes_from_triangles(triangles, volume_type=None):
    """BSphere{T}(p1,p2,p3) / BBox{T}(p1,p2,p3) for an (n, 3, 3) or (n, 9) CUDA tensor of triangles
    (bsphere.jl:43-112, bbox.jl:59-70)."""
    torch = _require_gpu()
    t = triangles.reshape(triangles.shape[0], 9).contiguous()
    vt = volume_type or BSphere(t.dtype)
    t = t.to(_torch_float(vt.flt))
    out = torch.empty((t.shape[0], abi.volume_width(vt.kind)), dtype=t.dtype, device="cuda")
    lib.call("ibvh_volumes_from_triangles", vt.kind, vt.flt, _ptr(t), t.shape[0], _ptr(out), _stream())
    return out


def load_obj_triangles(path, dtype=None):
    """Minimal Wavefront OBJ reader (`v x y z` / `f a b c ...`, 1-based or negative indices, `a/b/c` forms,
    polygons fan-triangulated) -> (n, 3, 3) CUDA tensor of triangles, the input of
    bounding_volumes_from_triangles.  Stands in for MeshIO/FileIO in the reference's benchmark scripts
    (benchmark/bvh_contact.jl:30-36); parsing is host-side file I/O, not part of the hot path."""
    torch = _require_gpu()
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    fa = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    if len(fa) and (fa.min() < 0 or fa.max() >= len(v)):
        raise ValueError("OBJ face index out of range")
    tri = v[fa] if len(fa) else np.zeros((0, 3, 3))
    return torch.from_numpy(tri).to(device="cuda", dtype=dtype or torch.float32)


def generate_spheres(n, seed, first_index=0, origin=(0.0, 0.0, 0.0), extent=(1.0, 1.0, 1.0), r0=0.01):
    """Deterministic synthetic BSphere{Float32} cloud (counter-based SplitMix64), on device."""
    torch = _require_gpu()
    out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    o = (C.c_float * 3)(*origin)
    e = (C.c_float * 3)(*extent)
    lib.call("ibvh_generate_spheres_f32", int(n), C.c_uint64(seed), int(first_index), o, e, C.c_float(r0), _ptr(out), _stream())
    return out
