"""ctypes binding of the CPU oracle (oracle/libibvh_oracle.so) on numpy arrays.

Test infrastructure only: the oracle is the checker for the HIP library, never a code path of the
product.  The `oracle_*` functions are host-pointer twins of the entry points in include/ibvh.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import implicitbvh_amd  # noqa: F401  (registers the package)
from implicitbvh_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libibvh_oracle.so")


def _load():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
    lib = C.CDLL(_SO)
    lib.oracle_brute_force_self.restype = C.c_int64
    lib.oracle_brute_force_pair.restype = C.c_int64
    lib.oracle_brute_force_rays.restype = C.c_int64
    lib.oracle_count_bad_ray_hits.restype = C.c_int64
    lib.oracle_count_bad_contacts.restype = C.c_int64
    lib.oracle_morton_split3_u16.restype = C.c_uint16
    lib.oracle_morton_split3_u32.restype = C.c_uint32
    lib.oracle_morton_split3_u64.restype = C.c_uint64
    lib.oracle_morton_split3_u16.argtypes = [C.c_uint16]
    lib.oracle_morton_split3_u32.argtypes = [C.c_uint32]
    lib.oracle_morton_split3_u64.argtypes = [C.c_uint64]
    return lib


lib = _load()


def _p(a):
    if a is None:
        return C.c_void_p(0)
    return C.c_void_p(a.ctypes.data)


# ---------------------------------------------------------------------------------------------
# tree
# ---------------------------------------------------------------------------------------------
def tree_shape(n):
    t = abi.Tree()
    abi.check(lib.oracle_tree_shape(C.c_int64(n), C.byref(t)), "oracle_tree_shape")
    return t


def compute_skips(tree):
    out = np.zeros(tree.levels, np.int64)
    abi.check(lib.oracle_compute_skips(C.byref(tree), _p(out)))
    return out


def memory_index(tree, i):
    out = C.c_int64()
    abi.check(lib.oracle_memory_index(C.byref(tree), C.c_int64(i), C.byref(out)))
    return out.value


def level_indices(tree, level):
    a, b = C.c_int64(), C.c_int64()
    abi.check(lib.oracle_level_indices(C.byref(tree), C.c_int64(level), C.byref(a), C.byref(b)))
    return a.value, b.value


def isvirtual(tree, i):
    out = C.c_int32()
    abi.check(lib.oracle_isvirtual(C.byref(tree), C.c_int64(i), C.byref(out)))
    return bool(out.value)


def compute_build_level(tree, frac):
    out = C.c_int64()
    abi.check(lib.oracle_compute_build_level(C.byref(tree), C.c_double(frac), C.byref(out)))
    return out.value


def layout_of(types):
    lay = abi.Layout()
    abi.check(lib.oracle_layout_of(C.byref(types), C.byref(lay)))
    return lay


# ---------------------------------------------------------------------------------------------
# geometry single-shots
# ---------------------------------------------------------------------------------------------
def as_volumes(arr, kind, flt):
    """(n, 4|6) numbers -> structured volume array."""
    a = np.ascontiguousarray(np.asarray(arr, dtype=abi.FLOAT_DTYPES[flt]).reshape(-1, abi.volume_width(kind)))
    return a.view(abi.volume_dtype(kind, flt)).reshape(-1)


def iscontact(kind_a, flt_a, a, kind_b, flt_b, b):
    va, vb = as_volumes(a, kind_a, flt_a), as_volumes(b, kind_b, flt_b)
    out = C.c_int32()
    abi.check(lib.oracle_iscontact(kind_a, flt_a, _p(va), kind_b, flt_b, _p(vb), C.byref(out)))
    return bool(out.value)


def isintersection(kind, flt, vol, p, d):
    v = as_volumes(vol, kind, flt)
    t = abi.FLOAT_DTYPES[flt]
    pp, dd = np.asarray(p, t).copy(), np.asarray(d, t).copy()
    out = C.c_int32()
    abi.check(lib.oracle_isintersection(kind, flt, _p(v), _p(pp), _p(dd), C.byref(out)))
    return bool(out.value)


def merge(types, a, b=None):
    va = as_volumes(a, types.leaf_kind, types.leaf_float)
    vb = None if b is None else as_volumes(b, types.leaf_kind, types.leaf_float)
    out = np.zeros(1, abi.node_dtype(types))
    abi.check(lib.oracle_merge(C.byref(types), _p(va), _p(vb), _p(out)))
    return out[0]


def volumes_from_triangles(kind, flt, tris):
    t = np.ascontiguousarray(np.asarray(tris, abi.FLOAT_DTYPES[flt]).reshape(-1, 9))
    out = np.zeros(len(t), abi.volume_dtype(kind, flt))
    abi.check(lib.oracle_volumes_from_triangles(kind, flt, _p(t), C.c_int64(len(t)), _p(out)))
    return out


def morton_split3(v, bits):
    f = {16: lib.oracle_morton_split3_u16, 32: lib.oracle_morton_split3_u32, 64: lib.oracle_morton_split3_u64}[bits]
    return int(f(v))


# ---------------------------------------------------------------------------------------------
# build
# ---------------------------------------------------------------------------------------------
class HostBVH:
    """A BVH in host memory (numpy), the oracle-side analogue of the product's BVH."""

    def __init__(self, types, tree, built_level, leaves, nodes, skips, extrema):
        self.types, self.tree, self.built_level = types, tree, built_level
        self.leaves, self.nodes, self.skips, self.extrema = leaves, nodes, skips, extrema

    def struct(self):
        b = abi.Bvh()
        b.types = self.types
        b.tree = self.tree
        b.built_level = self.built_level
        b.leaves = self.leaves.ctypes.data
        b.nodes = self.nodes.ctypes.data if len(self.nodes) else 0
        b.skips = self.skips.ctypes.data
        return b


def build(volumes, types, built_level=1, indices=None, compute_extrema=True, mins=None, maxs=None):
    """volumes: (n, 4|6) array.  indices != None -> pre-wrapped records with user indices."""
    vols = as_volumes(volumes, types.leaf_kind, types.leaf_float)
    n = len(vols)
    tree = tree_shape(n)
    d = abi.BuildDesc()
    d.types = types
    d.n = n
    d.built_level = built_level
    d.compute_extrema = 1 if compute_extrema else 0
    if mins is not None:
        d.mins[:] = list(map(float, mins))
        d.maxs[:] = list(map(float, maxs))
    leaves = np.zeros(n, abi.leaf_dtype(types))
    if indices is not None:
        leaves["volume"] = vols
        leaves["index"] = np.asarray(indices)
        d.already_wrapped = 1
    nodes = np.zeros(max(tree.real_nodes - tree.real_leaves, 0), abi.node_dtype(types))
    skips = np.zeros(tree.levels, abi.INDEX_DTYPES[types.index_type])
    ext = np.zeros(6, abi.FLOAT_DTYPES[types.leaf_float])
    abi.check(lib.oracle_build(C.byref(d), _p(vols), _p(leaves), _p(nodes), _p(skips), _p(ext)), "oracle_build")
    return HostBVH(types, tree, built_level, leaves, nodes, skips, ext)


def extrema(types, records, wrapped, expand=True):
    out = np.zeros(6, abi.FLOAT_DTYPES[types.leaf_float])
    abi.check(lib.oracle_extrema(C.byref(types), _p(records), int(wrapped), C.c_int64(len(records)), int(expand), _p(out)))
    return out


def morton_keys(types, records, wrapped, ext):
    out = np.zeros(len(records), abi.key_dtype(types))
    abi.check(lib.oracle_morton_keys(C.byref(types), _p(records), int(wrapped), C.c_int64(len(records)), _p(ext), _p(out)))
    return out


def sort_pairs(keys, vals):
    k, v = keys.copy(), vals.astype(np.uint32).copy()
    abi.check(lib.oracle_sort_pairs(k.dtype.itemsize, C.c_int64(len(k)), _p(k), _p(v)))
    return k, v


def aggregate(types, tree, built_level, leaves):
    nodes = np.zeros(max(tree.real_nodes - tree.real_leaves, 0), abi.node_dtype(types))
    abi.check(lib.oracle_aggregate(C.byref(types), C.byref(tree), C.c_int64(built_level), _p(leaves), _p(nodes)))
    return nodes


# ---------------------------------------------------------------------------------------------
# traversals
# ---------------------------------------------------------------------------------------------
def traverse_lvt(bvh, start_level=None, narrow=0):
    if start_level is None:
        start_level = max(1, bvh.built_level)
    s = bvh.struct()
    n = bvh.tree.real_leaves
    counts = np.zeros(n, abi.INDEX_DTYPES[bvh.types.index_type])
    total = C.c_int64()
    abi.check(lib.oracle_traverse_lvt_count(C.byref(s), C.c_int64(start_level), narrow, _p(counts), C.byref(total)))
    contacts = np.zeros(total.value, abi.pair_dtype(bvh.types))
    if total.value:
        abi.check(lib.oracle_traverse_lvt_write(C.byref(s), C.c_int64(start_level), narrow, _p(counts), _p(contacts)))
    return contacts, counts


def traverse_pair_lvt(bvh1, bvh2, start_level1=None, start_level2=None, narrow=0):
    sl1 = max(1, bvh1.built_level) if start_level1 is None else start_level1
    sl2 = max(1, bvh2.built_level) if start_level2 is None else start_level2
    s1, s2 = bvh1.struct(), bvh2.struct()
    n = max(bvh1.tree.real_leaves, bvh2.tree.real_leaves)
    counts = np.zeros(n, abi.INDEX_DTYPES[bvh1.types.index_type])
    total = C.c_int64()
    abi.check(lib.oracle_traverse_pair_lvt_count(C.byref(s1), C.byref(s2), C.c_int64(sl1), C.c_int64(sl2), narrow,
                                                 _p(counts), C.byref(total)))
    contacts = np.zeros(total.value, abi.pair_dtype(bvh1.types))
    if total.value:
        abi.check(lib.oracle_traverse_pair_lvt_write(C.byref(s1), C.byref(s2), C.c_int64(sl1), C.c_int64(sl2), narrow,
                                                     _p(counts), _p(contacts)))
    return contacts, counts


def _rays(bvh, points, directions):
    t = abi.FLOAT_DTYPES[bvh.types.leaf_float]
    p = np.ascontiguousarray(np.asarray(points, t).reshape(-1, 3))  # row i = ray i = Julia column i
    d = np.ascontiguousarray(np.asarray(directions, t).reshape(-1, 3))
    assert p.shape == d.shape
    return p, d


def traverse_rays_lvt(bvh, points, directions, start_level=1):
    p, d = _rays(bvh, points, directions)
    s = bvh.struct()
    nr = len(p)
    counts = np.zeros(nr, abi.INDEX_DTYPES[bvh.types.index_type])
    total = C.c_int64()
    abi.check(lib.oracle_traverse_rays_lvt_count(C.byref(s), _p(p), _p(d), C.c_int64(nr), C.c_int64(start_level),
                                                 _p(counts), C.byref(total)))
    contacts = np.zeros(total.value, abi.pair_dtype(bvh.types))
    if total.value:
        abi.check(lib.oracle_traverse_rays_lvt_write(C.byref(s), _p(p), _p(d), C.c_int64(nr), C.c_int64(start_level),
                                                     _p(counts), _p(contacts)))
    return contacts, counts


def _bfs_call(fn, types, *args):
    cap = 1024
    while True:
        buf = np.zeros(cap, abi.pair_dtype(types))
        res = abi.BfsResult()
        st = fn(*args, _p(buf), C.c_void_p(0), C.c_int64(cap), C.byref(res))
        if st == abi.ERR_CAPACITY:
            cap = int(res.required_capacity)
            continue
        abi.check(st)
        return buf[:res.num_contacts].copy(), res


def traverse_bfs(bvh, start_level=None, narrow=0):
    if start_level is None:
        start_level = max(bvh.tree.levels // 2, bvh.built_level)
    s = bvh.struct()
    return _bfs_call(lib.oracle_traverse_bfs, bvh.types, C.byref(s), C.c_int64(start_level), narrow)


def traverse_pair_bfs(bvh1, bvh2, start_level1=None, start_level2=None, narrow=0):
    sl1 = max(bvh1.tree.levels // 2, bvh1.built_level) if start_level1 is None else start_level1
    sl2 = max(bvh2.tree.levels // 2, bvh2.built_level) if start_level2 is None else start_level2
    s1, s2 = bvh1.struct(), bvh2.struct()
    return _bfs_call(lib.oracle_traverse_pair_bfs, bvh1.types, C.byref(s1), C.byref(s2), C.c_int64(sl1),
                     C.c_int64(sl2), narrow)


def traverse_rays_bfs(bvh, points, directions, start_level=1):
    p, d = _rays(bvh, points, directions)
    s = bvh.struct()
    return _bfs_call(lib.oracle_traverse_rays_bfs, bvh.types, C.byref(s), _p(p), _p(d), C.c_int64(len(p)),
                     C.c_int64(start_level))


# ---------------------------------------------------------------------------------------------
# brute force + generators
# ---------------------------------------------------------------------------------------------
def brute_force_self(kind, flt, volumes):
    v = as_volumes(volumes, kind, flt)
    cap = 1 << 16
    while True:
        out = np.zeros((cap, 2), np.int64)
        c = lib.oracle_brute_force_self(kind, flt, _p(v), C.c_int64(len(v)), _p(out), C.c_int64(cap))
        if c <= cap:
            return out[:c]
        cap = c


def brute_force_pair(kind, flt, va, vb):
    a, b = as_volumes(va, kind, flt), as_volumes(vb, kind, flt)
    cap = 1 << 16
    while True:
        out = np.zeros((cap, 2), np.int64)
        c = lib.oracle_brute_force_pair(kind, flt, _p(a), C.c_int64(len(a)), _p(b), C.c_int64(len(b)), _p(out),
                                        C.c_int64(cap))
        if c <= cap:
            return out[:c]
        cap = c


def brute_force_rays(kind, flt, volumes, points, directions):
    v = as_volumes(volumes, kind, flt)
    t = abi.FLOAT_DTYPES[flt]
    p = np.ascontiguousarray(np.asarray(points, t).reshape(-1, 3))
    d = np.ascontiguousarray(np.asarray(directions, t).reshape(-1, 3))
    cap = 1 << 16
    while True:
        out = np.zeros((cap, 2), np.int64)
        c = lib.oracle_brute_force_rays(kind, flt, _p(v), C.c_int64(len(v)), _p(p), _p(d), C.c_int64(len(p)),
                                        _p(out), C.c_int64(cap))
        if c <= cap:
            return out[:c]
        cap = c


def count_bad_ray_hits(kind, flt, volumes, points, directions, pairs):
    v = as_volumes(volumes, kind, flt)
    t = abi.FLOAT_DTYPES[flt]
    p = np.ascontiguousarray(np.asarray(points, t).reshape(-1, 3))
    d = np.ascontiguousarray(np.asarray(directions, t).reshape(-1, 3))
    pr = np.ascontiguousarray(np.asarray(pairs, np.int64).reshape(-1, 2))
    return int(lib.oracle_count_bad_ray_hits(kind, flt, _p(v), _p(p), _p(d), _p(pr), C.c_int64(len(pr))))


def count_bad_contacts(kind, flt, va, vb, pairs):
    a, b = as_volumes(va, kind, flt), as_volumes(vb, kind, flt)
    pr = np.ascontiguousarray(np.asarray(pairs, np.int64).reshape(-1, 2))
    return int(lib.oracle_count_bad_contacts(kind, flt, _p(a), _p(b), _p(pr), C.c_int64(len(pr))))


def generate_spheres_f32(n, seed, first_index=0, origin=(0, 0, 0), extent=(1, 1, 1), r0=0.01):
    out = np.zeros((n, 4), np.float32)
    o = (C.c_float * 3)(*origin)
    e = (C.c_float * 3)(*extent)
    abi.check(lib.oracle_generate_spheres_f32(C.c_int64(n), C.c_uint64(seed), C.c_int64(first_index), o, e,
                                              C.c_float(r0), _p(out)))
    return out


def load_native():
    """The -O3 -march=native build of the same source (oracle/Makefile target `native`), made on the machine that runs it:
    bench.py's timed CPU baseline only.  Returns None if it cannot be built here."""
    import subprocess
    so = os.path.join(_ROOT, "oracle", "_native", "libibvh_oracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=600)
        nat = C.CDLL(so)
        for name in ("oracle_bench_build_traverse_f32", "oracle_bench_pair_lvt", "oracle_bench_rays_lvt", "oracle_lvt_test_counts"):
            getattr(nat, name).restype = C.c_int
        return nat
    except Exception:
        return None


def bench_build_traverse_f32(volumes, threads, native=None):
    """Timed multi-threaded CPU restatement on BSphere{F32}/BBox{F32}/I32/U32."""
    types = abi.make_types()
    v = as_volumes(volumes, abi.BSPHERE, abi.F32)
    n = len(v)
    tree = tree_shape(n)
    leaves = np.zeros(n, abi.leaf_dtype(types))
    nodes = np.zeros(max(tree.real_nodes - tree.real_leaves, 0), abi.node_dtype(types))
    skips = np.zeros(tree.levels, np.int32)
    counts = np.zeros(n, np.int32)
    cap = max(16 * n, 1024)
    contacts = np.zeros(cap, abi.pair_dtype(types))
    nc, tb, tt = C.c_int64(), C.c_double(), C.c_double()
    st = (native or lib).oracle_bench_build_traverse_f32(_p(v), C.c_int64(n), int(threads), _p(leaves), _p(nodes), _p(skips),
                                             _p(counts), _p(contacts), C.c_int64(cap), C.byref(nc), C.byref(tb),
                                             C.byref(tt))
    abi.check(st)
    bvh = HostBVH(types, tree, 1, leaves, nodes, skips, None)
    return bvh, contacts[:nc.value], tb.value, tt.value


def bench_pair_lvt(bvh1, bvh2, threads, native=None, capacity=None):
    """Timed two-pass LVT pair traversal of two pre-built host BVHs on `threads` OpenMP threads (the cpu_baseline leg of
    config 4; protocol of benchmark/bvh_contact_pair.jl:38-46).  Returns (num_contacts, seconds)."""
    s1, s2 = bvh1.struct(), bvh2.struct()
    n = max(bvh1.tree.real_leaves, bvh2.tree.real_leaves)
    counts = np.zeros(n, abi.INDEX_DTYPES[bvh1.types.index_type])
    cap = int(capacity or max(4 * n, 1024))
    while True:
        contacts = np.zeros(cap, abi.pair_dtype(bvh1.types))
        nc, ts = C.c_int64(), C.c_double()
        st = (native or lib).oracle_bench_pair_lvt(C.byref(s1), C.byref(s2), C.c_int64(max(1, bvh1.built_level)),
                                                   C.c_int64(max(1, bvh2.built_level)), int(threads), _p(counts), _p(contacts),
                                                   C.c_int64(cap), C.byref(nc), C.byref(ts))
        if st == abi.ERR_CAPACITY:
            cap = nc.value
            continue
        abi.check(st)
        return nc.value, ts.value


def bench_rays_lvt(bvh, points, directions, threads, native=None, capacity=None):
    """Timed two-pass LVT ray traversal on `threads` OpenMP threads (cpu_baseline leg of config 3; protocol of
    benchmark/bvh_rays.jl:36-58).  Returns (num_hits, seconds)."""
    p, d = _rays(bvh, points, directions)
    s = bvh.struct()
    counts = np.zeros(len(p), abi.INDEX_DTYPES[bvh.types.index_type])
    cap = int(capacity or max(32 * len(p), 1024))
    while True:
        contacts = np.zeros(cap, abi.pair_dtype(bvh.types))
        nc, ts = C.c_int64(), C.c_double()
        st = (native or lib).oracle_bench_rays_lvt(C.byref(s), _p(p), _p(d), C.c_int64(len(p)), C.c_int64(1), int(threads),
                                                   _p(counts), _p(contacts), C.c_int64(cap), C.byref(nc), C.byref(ts))
        if st == abi.ERR_CAPACITY:
            cap = nc.value
            continue
        abi.check(st)
        return nc.value, ts.value


def lvt_test_counts(bvh, bvh2=None, points=None, directions=None, threads=1, start_level=None, start_level2=None, native=None):
    """Work of the reference's leaf-vs-tree walk (one pass): (node tests, leaf tests, contacts) — SURVEY.md §8d
    "touched bytes" = 24 B x (node tests + leaf tests) for the bench types."""
    s = bvh.struct()
    sl1 = max(1, bvh.built_level) if start_level is None else start_level
    nt, lt, nc = C.c_int64(), C.c_int64(), C.c_int64()
    idt = abi.INDEX_DTYPES[bvh.types.index_type]
    f = (native or lib).oracle_lvt_test_counts
    if points is not None:
        p, d = _rays(bvh, points, directions)
        counts = np.zeros(len(p), idt)
        st = f(C.byref(s), C.c_void_p(0), _p(p), _p(d), C.c_int64(len(p)), C.c_int64(sl1 if start_level is not None else 1),
               C.c_int64(0), int(threads), _p(counts), C.byref(nt), C.byref(lt), C.byref(nc))
    elif bvh2 is not None:
        s2 = bvh2.struct()
        sl2 = max(1, bvh2.built_level) if start_level2 is None else start_level2
        counts = np.zeros(max(bvh.tree.real_leaves, bvh2.tree.real_leaves), idt)
        st = f(C.byref(s), C.byref(s2), C.c_void_p(0), C.c_void_p(0), C.c_int64(0), C.c_int64(sl1), C.c_int64(sl2), int(threads),
               _p(counts), C.byref(nt), C.byref(lt), C.byref(nc))
    else:
        counts = np.zeros(1, idt)
        st = f(C.byref(s), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_int64(0), C.c_int64(sl1), C.c_int64(0), int(threads),
               _p(counts), C.byref(nt), C.byref(lt), C.byref(nc))
    abi.check(st)
    return nt.value, lt.value, nc.value


def pairs_as_tuples(contacts):
    """structured IndexPair array or (n,2) array -> list of (a, b) python tuples."""
    if contacts.dtype.names:
        return list(zip(contacts["a"].tolist(), contacts["b"].tolist()))
    return [tuple(r) for r in contacts.tolist()]
