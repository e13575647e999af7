"""CPU-side checks of the product: the C-ABI library loads and exports every symbol that
include/ibvh.h declares, the host shape math matches the oracle, and the Python mirror validates
arguments like the reference.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as orc
import implicitbvh_amd as ibvh
from implicitbvh_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ibvh.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ibvh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    l = lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(l, s), f"libibvh.so does not export {s}"
    # and the binding table covers the header exactly
    assert sorted(lib.SIGNATURES) == syms


def test_version_and_status_strings():
    l = lib.load()
    assert b"gfx950" in l.ibvh_version()
    assert l.ibvh_status_string(abi.ERR_DOMAIN).startswith(b"domain")


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 11, 100, 1000, 249882, 10**6, 10**7, 10**8, 2**20, 2**20 + 1])
def test_tree_shape_and_index_math_match_oracle(n):
    t = ibvh.ImplicitTree(n)
    o = orc.tree_shape(n)
    assert (t.levels, t.real_leaves, t.real_nodes, t.virtual_leaves, t.virtual_nodes) == o.astuple()
    sk = (C.c_int64 * t.levels)()
    lib.call("ibvh_compute_skips", C.byref(t._t), sk)
    assert list(sk) == orc.compute_skips(o).tolist()
    rng = np.random.default_rng(n)
    for level in range(1, t.levels + 1):
        assert ibvh.level_indices(t, level) == orc.level_indices(o, level)
    for idx in {1, 2 ** t.levels - 1, *rng.integers(1, 2 ** t.levels, 50).tolist()}:
        assert ibvh.memory_index(t, idx) == orc.memory_index(o, idx)
        assert ibvh.isvirtual(t, idx) == orc.isvirtual(o, idx)
    for frac in (0.0, 0.25, 0.5, 0.75, 1.0):
        out = C.c_int64()
        lib.call("ibvh_compute_build_level", C.byref(t._t), frac, C.byref(out))
        assert out.value == orc.compute_build_level(o, frac)


def test_tree_errors():
    with pytest.raises(abi.DomainError):
        ibvh.ImplicitTree(0)
    t = ibvh.ImplicitTree(5)
    with pytest.raises(IndexError):
        ibvh.memory_index(t, 16)
    with pytest.raises(IndexError):
        ibvh.level_indices(t, 5)
    with pytest.raises(IndexError):
        ibvh.isvirtual(t, 0)


def test_layouts_match_oracle_and_numpy():
    for lk in (abi.BSPHERE, abi.BBOX):
        for lf in (abi.F32, abi.F64):
            for nk in (abi.BSPHERE, abi.BBOX):
                for nf in (abi.F32, abi.F64):
                    for it in (abi.I32, abi.I64):
                        for mt in (abi.U16, abi.U32, abi.U64):
                            t = abi.make_types(lk, lf, nk, nf, it, mt)
                            lay = abi.Layout()
                            st = lib.load().ibvh_layout_of(C.byref(t), C.byref(lay))
                            if not abi.combo_supported(t):
                                assert st == abi.ERR_UNSUPPORTED
                                continue
                            assert st == abi.OK
                            o = orc.layout_of(t)
                            for f, _ in abi.Layout._fields_:
                                assert getattr(lay, f) == getattr(o, f)
                            assert abi.leaf_dtype(t).itemsize == lay.leaf_bytes
                            assert lay.leaf_bytes % 8 == 0  # volumes move as 8-byte words (ibvh_common.hpp)


def test_scratch_queries():
    t = abi.make_types()
    need = C.c_size_t()
    lib.call("ibvh_build_scratch_bytes", C.byref(t), 10**6, C.byref(need))
    assert 10**6 * (4 + 4 + 4 + 4) <= need.value <= 10**6 * 64
    with pytest.raises(abi.DomainError):
        lib.call("ibvh_build_scratch_bytes", C.byref(t), 0, C.byref(need))
    lib.call("ibvh_lvt_scratch_bytes", C.byref(t), 10**6, 8, C.byref(need))
    assert need.value >= 8
    lib.call("ibvh_bfs_counters_bytes", 21, C.byref(need))
    assert need.value >= 8 * 22


def test_options_validation_like_argcheck():
    ibvh.BVHOptions()
    for bad in ({"num_threads": 0}, {"block_size": 0}, {"min_sorts_per_thread": -1}):
        with pytest.raises(ValueError):
            ibvh.BVHOptions(**bad)
    with pytest.raises(ValueError):
        ibvh.DefaultMortonAlgorithm(np.uint8)
    assert ibvh.BVHOptions(index=np.int64, morton=ibvh.DefaultMortonAlgorithm(np.uint64)).morton_code == abi.U64


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ibvh.BVH(torch.zeros((4, 4)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ibvh.generate_spheres(10, 1)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "implicitbvh.jl_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".jl")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle/" not in text and "ibvh_oracle" not in text and "oracle_lib" not in text, f
