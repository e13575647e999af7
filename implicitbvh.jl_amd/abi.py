"""ctypes mirror of include/ibvh.h: enums, POD descriptors, record dtypes, status -> exception.

Record layouts follow Julia's isbits struct layout (= C layout), see include/ibvh.h:
BSphere{T} (bsphere.jl:26-29), BBox{T} (bbox.jl:35-38), BoundingVolume{V,I,M}
(bounding_volumes.jl:55-59), IndexPair{I} (traverse.jl:6).
"""
import ctypes as C

import numpy as np

ABI_VERSION = 5  # IBVH_ABI_VERSION of the include/ibvh.h this mirror was written against

# enums ---------------------------------------------------------------------------------------
BSPHERE, BBOX = 0, 1
F32, F64 = 0, 1
I32, I64 = 0, 1
U16, U32, U64 = 0, 1, 2
NARROW_NONE, NARROW_MORTON_LT, NARROW_INDEX_LT, NARROW_RAY_ORIGIN_OUTSIDE = 0, 1, 2, 3
NARROW_MASK, OUTPUT_POSITIONS = 0xff, 0x100  # IBVH_NARROW_MASK, IBVH_OUTPUT_POSITIONS
PAIR_SMALLER_DRIVES = 0x200  # IBVH_PAIR_SMALLER_DRIVES: pair LVT traversals, the BVH with fewer leaves supplies the work items (a set, not the reference's order)

OK, ERR_INVALID_ARG, ERR_DOMAIN, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_OVERFLOW, ERR_HIP, ERR_SCRATCH, ERR_PEER = range(9)

FLOAT_DTYPES = {F32: np.float32, F64: np.float64}
INDEX_DTYPES = {I32: np.int32, I64: np.int64}
MORTON_DTYPES = {U16: np.uint16, U32: np.uint32, U64: np.uint64}
MORTON_BITS = {U16: 15, U32: 30, U64: 63}  # 3 x (5, 10, 21), morton/default.jl:167-169


class DomainError(ValueError):
    """Julia's DomainError (implicit_tree.jl:78-80)."""


class CapacityError(RuntimeError):
    def __init__(self, msg, required):
        super().__init__(msg)
        self.required = required


class Types(C.Structure):
    _fields_ = [("leaf_kind", C.c_int32), ("leaf_float", C.c_int32), ("node_kind", C.c_int32),
                ("node_float", C.c_int32), ("index_type", C.c_int32), ("morton_type", C.c_int32)]

    def key(self):
        return (self.leaf_kind, self.leaf_float, self.node_kind, self.node_float, self.index_type,
                self.morton_type)


class Tree(C.Structure):
    _fields_ = [("levels", C.c_int64), ("real_leaves", C.c_int64), ("real_nodes", C.c_int64),
                ("virtual_leaves", C.c_int64), ("virtual_nodes", C.c_int64)]

    def astuple(self):
        return (self.levels, self.real_leaves, self.real_nodes, self.virtual_leaves, self.virtual_nodes)


class Layout(C.Structure):
    _fields_ = [("volume_bytes", C.c_int64), ("node_bytes", C.c_int64), ("index_off", C.c_int64),
                ("morton_off", C.c_int64), ("leaf_bytes", C.c_int64), ("pair_bytes", C.c_int64)]


class Bvh(C.Structure):
    _fields_ = [("types", Types), ("tree", Tree), ("built_level", C.c_int64), ("leaves", C.c_void_p),
                ("nodes", C.c_void_p), ("skips", C.c_void_p)]


MAX_SORT_LEVELS = 4  # IBVH_MAX_SORT_LEVELS (include/ibvh.h)


class BuildDesc(C.Structure):
    _fields_ = [("types", Types), ("n", C.c_int64), ("built_level", C.c_int64),
                ("already_wrapped", C.c_int32), ("compute_extrema", C.c_int32),
                ("mins", C.c_double * 3), ("maxs", C.c_double * 3),
                ("sort_levels", C.c_int32), ("sort_equalize", C.c_int32), ("skew_flag", C.c_void_p)]


class BfsResult(C.Structure):
    _fields_ = [("num_contacts", C.c_int64), ("num_checks", C.c_int64), ("contacts_in", C.c_int64),
                ("required_capacity", C.c_int64), ("resume_step", C.c_int64), ("resume_num", C.c_int64)]


# ---- multi-GPU build (include/ibvh.h, "multi-GPU build: the driver") ------------------------------------------
DIST_MAX_RANKS = 256
COMM_F64, COMM_I64, COMM_I32 = 0, 1, 2
COMM_MAX, COMM_SUM, COMM_MIN = 0, 1, 2
COMM_ALL_REDUCE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)
COMM_ALL_GATHER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
COMM_ALL_TO_ALL_V = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_void_p)


class Comm(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("size", C.c_int32), ("all_reduce", COMM_ALL_REDUCE),
                ("all_gather", COMM_ALL_GATHER), ("all_to_all_v", COMM_ALL_TO_ALL_V)]


class SplitterSearch(C.Structure):
    _fields_ = [("size", C.c_int32), ("key_bits", C.c_int32), ("decided", C.c_int32), ("all_done", C.c_int32),
                ("next_bits", C.c_int32), ("next_shift", C.c_int32), ("num_rows", C.c_int32), ("reserved_", C.c_int32),
                ("n_global", C.c_int64), ("tolerance", C.c_double),
                ("rows", C.c_uint64 * DIST_MAX_RANKS), ("prefix", C.c_uint64 * DIST_MAX_RANKS),
                ("splitters", C.c_uint64 * DIST_MAX_RANKS), ("below", C.c_int64 * DIST_MAX_RANKS),
                ("row_of", C.c_int32 * DIST_MAX_RANKS), ("done", C.c_uint8 * DIST_MAX_RANKS)]


class DistPlan(C.Structure):
    _fields_ = [("size", C.c_int32), ("levels_used", C.c_int32), ("n_local", C.c_int64), ("n_global", C.c_int64),
                ("base", C.c_int64), ("n_slice", C.c_int64), ("record_bytes", C.c_int64), ("extrema", C.c_double * 6),
                ("splitters", C.c_uint64 * DIST_MAX_RANKS), ("send_counts", C.c_int64 * DIST_MAX_RANKS),
                ("recv_counts", C.c_int64 * DIST_MAX_RANKS)]


DIST_CROSS_BOXES = 16


class DistCrossPlan(C.Structure):
    """ibvh_dist_cross_plan_t (include/ibvh.h)"""
    _fields_ = [("size", C.c_int32), ("rank", C.c_int32), ("n_recv", C.c_int32), ("cache_slots", C.c_int32),
                ("import_bytes", C.c_int64), ("scratch_bytes", C.c_int64), ("export_bytes", C.c_int64), ("build_offset", C.c_int64),
                ("recv_rank", C.c_int32 * DIST_MAX_RANKS), ("recv_leaves", C.c_int64 * DIST_MAX_RANKS),
                ("recv_offset", C.c_int64 * DIST_MAX_RANKS), ("scratch_offset", C.c_int64 * DIST_MAX_RANKS),
                ("slice_leaves", C.c_int64 * DIST_MAX_RANKS), ("touches", C.c_int32 * DIST_MAX_RANKS),
                ("send_leaves", C.c_int64 * DIST_MAX_RANKS), ("send_offset", C.c_int64 * DIST_MAX_RANKS),
                ("n_boxes", C.c_int32 * DIST_MAX_RANKS), ("boxes", ((C.c_double * 6) * DIST_CROSS_BOXES) * DIST_MAX_RANKS)]


def dist_cross_scratch(size):
    """IBVH_DIST_CROSS_SCRATCH(size) of include/ibvh.h"""
    return (DIST_CROSS_BOXES * 48 + 16) * (size + 1) + 16 * size + 512


def volume_dtype(kind, flt):
    t = FLOAT_DTYPES[flt]
    if kind == BSPHERE:
        return np.dtype([("x", t, (3,)), ("r", t)], align=True)
    return np.dtype([("lo", t, (3,)), ("up", t, (3,))], align=True)


def volume_width(kind):
    """Number of scalars in a volume: BSphere 4 (x, r), BBox 6 (lo, up)."""
    return 4 if kind == BSPHERE else 6


def leaf_dtype(types):
    return np.dtype([("volume", volume_dtype(types.leaf_kind, types.leaf_float)),
                     ("index", INDEX_DTYPES[types.index_type]),
                     ("morton", MORTON_DTYPES[types.morton_type])], align=True)


def node_dtype(types):
    return volume_dtype(types.node_kind, types.node_float)


def pair_dtype(types):
    t = INDEX_DTYPES[types.index_type]
    return np.dtype([("a", t), ("b", t)], align=True)


def key_dtype(types):
    return np.uint64 if types.morton_type == U64 else np.uint32


def make_types(leaf_kind=BSPHERE, leaf_float=F32, node_kind=BBOX, node_float=F32, index_type=I32,
               morton_type=U32):
    return Types(leaf_kind, leaf_float, node_kind, node_float, index_type, morton_type)


def combo_supported(t):
    """NodeType(leaf) must exist in the reference (merge.jl); any node float type (build.jl:198-205)."""
    if t.node_kind == BSPHERE and t.leaf_kind != BSPHERE:
        return False
    return True


_STATUS_TEXT = {
    ERR_INVALID_ARG: "invalid argument (ArgumentError in the reference)",
    ERR_DOMAIN: "must have at least one geometry! (DomainError)",
    ERR_UNSUPPORTED: "type combination not supported by libibvh",
    ERR_CAPACITY: "caller buffer too small",
    ERR_OVERFLOW: "count does not fit the index type",
    ERR_HIP: "HIP runtime error",
    ERR_SCRATCH: "scratch buffer too small",
    ERR_PEER: "another rank's arguments were not acceptable (every rank returned together; that rank reports its own error)",
}


def check(status, what="ibvh"):
    """Map an ibvh_status to the exception the Julia shim raises for it."""
    if status == OK:
        return
    msg = f"{what}: {_STATUS_TEXT.get(status, 'status %d' % status)}"
    if status == ERR_DOMAIN:
        raise DomainError(msg)
    if status in (ERR_INVALID_ARG, ERR_UNSUPPORTED):
        raise ValueError(msg)
    if status == ERR_OVERFLOW:
        raise OverflowError(msg)
    raise RuntimeError(msg)
