"""ctypes binding of libibvh.so (the HIP library; C ABI in include/ibvh.h).

There is no fallback: if the shared library is missing this raises, loudly, and every product
operation fails.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C implicitbvh.jl_amd/csrc`.
"""
import ctypes as C
import os

from . import abi

_PKG = os.path.dirname(os.path.abspath(__file__))
# IBVH_LIB: load another build of the same library (diagnostic builds, tools/phase_stamps.sh); default: the in-tree one
LIB_PATH = os.environ.get("IBVH_LIB") or os.path.join(_PKG, "libibvh.so")

_vp, _i64, _i32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
_P = C.POINTER

# name -> argtypes (restype is always ibvh_status = int unless listed in _RESTYPES)
SIGNATURES = {
    "ibvh_tree_shape": [_i64, _P(abi.Tree)],
    "ibvh_compute_skips": [_P(abi.Tree), _P(_i64)],
    "ibvh_memory_index": [_P(abi.Tree), _i64, _P(_i64)],
    "ibvh_level_indices": [_P(abi.Tree), _i64, _P(_i64), _P(_i64)],
    "ibvh_isvirtual": [_P(abi.Tree), _i64, _P(_i32)],
    "ibvh_compute_build_level": [_P(abi.Tree), C.c_double, _P(_i64)],
    "ibvh_layout_of": [_P(abi.Types), _P(abi.Layout)],
    "ibvh_build_scratch_bytes": [_P(abi.Types), _i64, _P(_sz)],
    "ibvh_build": [_P(abi.BuildDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "ibvh_extrema": [_P(abi.Types), _vp, _i32, _i64, _i32, _vp, _vp, _sz, _vp],
    "ibvh_morton_keys": [_P(abi.Types), _vp, _i32, _i64, _vp, _vp, _vp],
    "ibvh_sort_pairs": [_i32, _i32, _i64, _vp, _vp, _vp, _vp, _P(_i32), _vp, _sz, _vp],
    "ibvh_sort_scratch_bytes": [_i32, _i64, _P(_sz)],
    "ibvh_aggregate": [_P(abi.Types), _P(abi.Tree), _i64, _vp, _vp, _vp],
    "ibvh_lvt_scratch_bytes": [_P(abi.Types), _i64, _i32, _P(_sz)],
    "ibvh_traverse_lvt_count": [_P(abi.Bvh), _i64, _i32, _vp, _P(_i64), _vp, _sz, _vp],
    "ibvh_traverse_lvt_write": [_P(abi.Bvh), _i64, _i32, _vp, _vp, _vp, _sz, _vp],
    "ibvh_traverse_pair_lvt_count": [_P(abi.Bvh), _P(abi.Bvh), _i64, _i64, _i32, _vp, _P(_i64), _vp, _sz, _vp],
    "ibvh_traverse_pair_lvt_write": [_P(abi.Bvh), _P(abi.Bvh), _i64, _i64, _i32, _vp, _vp, _vp, _sz, _vp],
    "ibvh_rays_scratch_bytes": [_P(abi.Bvh), _i64, _i32, _P(_sz)],
    "ibvh_traverse_rays_lvt_count": [_P(abi.Bvh), _vp, _vp, _i64, _i64, _i32, _vp, _P(_i64), _vp, _sz, _vp],
    "ibvh_traverse_rays_lvt_write": [_P(abi.Bvh), _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _sz, _vp],
    "ibvh_traverse_lvt_enqueue": [_P(abi.Bvh), _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp],
    "ibvh_traverse_pair_lvt_enqueue": [_P(abi.Bvh), _P(abi.Bvh), _i64, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp],
    "ibvh_traverse_rays_lvt_enqueue": [_P(abi.Bvh), _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp],
    "ibvh_lvt_total": [_vp, _P(_i64), _vp],
    "ibvh_bfs_initial_capacity": [_P(abi.Bvh), _i64, _P(_i64)],
    "ibvh_bfs_pair_initial_capacity": [_P(abi.Bvh), _P(abi.Bvh), _i64, _i64, _P(_i64)],
    "ibvh_bfs_rays_initial_capacity": [_P(abi.Bvh), _i64, _i64, _P(_i64)],
    "ibvh_bfs_counters_bytes": [_i64, _P(_sz)],
    "ibvh_traverse_bfs": [_P(abi.Bvh), _i64, _i32, _vp, _vp, _i64, _vp, _P(abi.BfsResult), _vp],
    "ibvh_traverse_pair_bfs": [_P(abi.Bvh), _P(abi.Bvh), _i64, _i64, _i32, _vp, _vp, _i64, _vp,
                               _P(abi.BfsResult), _vp],
    "ibvh_traverse_rays_bfs": [_P(abi.Bvh), _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _P(abi.BfsResult), _vp],
    "ibvh_expand_extrema": [_i32, _vp, _vp],
    "ibvh_dist_pack_extrema": [_i32, _vp, _i32, _i32, _i32, _i64, _vp, _vp],
    "ibvh_dist_unpack_extrema": [_i32, _vp, _vp, _vp],
    "ibvh_dist_partition_scratch_bytes": [_i64, _P(_sz)],
    "ibvh_dist_partition": [_i32, _vp, _i64, _P(C.c_uint64), _i32, _vp, _vp, _vp, _sz, _vp],
    "ibvh_key_histogram": [_i32, _vp, _i64, _i32, _i32, _i32, _P(C.c_uint64), _i32, _vp, _vp],
    "ibvh_pack_records": [_P(abi.Types), _vp, _vp, _vp, _i64, _i64, _vp, _vp],
    "ibvh_comm_from_rccl": [_vp, _i32, _i32, _P(abi.Comm)],
    "ibvh_splitter_search_init": [_P(abi.SplitterSearch), _i32, _i32, _i64, C.c_double],
    "ibvh_splitter_search_step": [_P(abi.SplitterSearch), _P(_i64)],
    "ibvh_dist_scratch_bytes": [_P(abi.Types), _i64, _i32, _P(_sz)],
    "ibvh_dist_plan": [_P(abi.Types), _P(abi.Comm), _vp, _i64, C.c_double, _vp, _sz, _P(abi.DistPlan), _vp],
    "ibvh_dist_exchange": [_P(abi.Types), _P(abi.Comm), _vp, _P(abi.DistPlan), _vp, _sz, _vp, _vp],
    "ibvh_dist_cross_plan": [_P(abi.Comm), _P(abi.Bvh), _i32, _vp, _sz, _P(abi.DistCrossPlan), _vp],
    "ibvh_dist_cross_exchange": [_P(abi.Comm), _P(abi.Bvh), _P(abi.DistCrossPlan), _vp, _vp, _vp, _sz, _vp],
    "ibvh_dist_cross_count": [_P(abi.Bvh), _P(abi.DistCrossPlan), _vp, _vp, _sz, _P(_i64), _P(_i64), _vp],
    "ibvh_dist_cross_write": [_P(abi.Bvh), _P(abi.DistCrossPlan), _vp, _vp, _sz, _P(_i64), _vp, _vp],
    "ibvh_comm_release": [_P(abi.Comm)],
    "ibvh_volumes_from_triangles": [_i32, _i32, _vp, _i64, _vp, _vp],
    "ibvh_generate_spheres_f32": [_i64, C.c_uint64, _i64, _P(C.c_float), _P(C.c_float), C.c_float, _vp, _vp],
    "ibvh_profile_enable": [_i32],
    "ibvh_profile_count": [_P(_i64)],
    "ibvh_profile_get": [_i64, _P(C.c_char_p), _P(C.c_float)],
    "ibvh_lvt_work_counters": [_P(abi.Bvh), _P(abi.Bvh), _vp, _vp, _i64, _vp, _vp, _vp],
    "ibvh_set_tuning": [C.c_char_p, _i32],
    "ibvh_get_tuning": [C.c_char_p, _P(_i32)],
    "ibvh_abi_version": [],
    "ibvh_version": [],
    "ibvh_status_string": [_i32],
}
_RESTYPES = {"ibvh_version": C.c_char_p, "ibvh_status_string": C.c_char_p, "ibvh_abi_version": C.c_int32}

_lib = None


def load():
    """Load libibvh.so once; raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built (no CPU fallback exists). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    got = lib.ibvh_abi_version()
    if got != abi.ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {got}, this binding was written against {abi.ABI_VERSION} "
                          "(include/ibvh.h IBVH_ABI_VERSION): rebuild the library")
    # Development knobs (measurement scripts only): IBVH_TUNING="name=value,name=value" is applied once, here, through
    # ibvh_set_tuning — the library itself never reads the environment.
    for item in filter(None, os.environ.get("IBVH_TUNING", "").split(",")):
        name, _, value = item.partition("=")
        abi.check(lib.ibvh_set_tuning(name.strip().encode(), int(value)), f"IBVH_TUNING {item!r}")
    _lib = lib
    return lib


def set_tuning(name, value):
    """ibvh_set_tuning: a process-wide development knob (include/ibvh.h).  Sizes the mirror has memoised (scratch bytes
    depend on the sort's geometry) are forgotten."""
    call("ibvh_set_tuning", name.encode(), int(value))
    import sys
    api = sys.modules.get(__package__ + ".api")
    if api is not None:
        api._shape_memo.clear()


def call(name, *args):
    """Call an entry point and raise the exception the Julia shim would for a non-zero status."""
    st = getattr(load(), name)(*args)
    abi.check(st, name)
    return st
