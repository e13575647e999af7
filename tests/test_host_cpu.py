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
    assert 10**6 * (4 + 4 + 4 + 4) <= need.value <= 10**6 * 96  # keys + positions, twice; two record stagings; tables
    with pytest.raises(abi.DomainError):
        lib.call("ibvh_build_scratch_bytes", C.byref(t), 0, C.byref(need))
    lib.call("ibvh_lvt_scratch_bytes", C.byref(t), 10**6, 8, C.byref(need))
    assert need.value >= 8
    lib.call("ibvh_bfs_counters_bytes", 21, C.byref(need))
    assert need.value >= 8 * 22


def _ray_scratch(n_leaves, n_rays, types=None, built_level=1):
    b = abi.Bvh()
    b.types = types or abi.make_types()
    lib.call("ibvh_tree_shape", int(n_leaves), C.byref(b.tree))
    b.built_level = built_level
    need, base = C.c_size_t(), C.c_size_t()
    lib.call("ibvh_rays_scratch_bytes", C.byref(b), int(n_rays), 8, C.byref(need))
    lib.call("ibvh_lvt_scratch_bytes", C.byref(b.types), int(n_rays), 8, C.byref(base))
    return need.value, base.value


def test_ray_scratch_follows_the_binned_path_rule():
    """ibvh_rays_scratch_bytes (host arithmetic only) makes room for the binned ray path exactly where the launch code takes it
    (csrc/ibvh_lvt.hip rays_bin_plan): trees (one float type throughout) of >= 17 levels, or >= 13 under small batches — 40 bytes x 16 items per
    ray plus tables; everything else gets the leaf-query scratch; the knobs move the rule."""
    need, base = _ray_scratch(7_201_012, 1_000_000)
    assert 16 * 40 * 10**6 <= need <= 16 * 40 * 10**6 + 64 * 2**20
    need, base = _ray_scratch(7_201_012, 30_000)          # few rays: still binned (subtrees nobody reaches are never loaded)
    assert need >= 16 * 40 * 30_000
    need, base = _ray_scratch(7_201_012, 64)              # a handful of rays: the walk is a dependent chain, cutting it pays most
    assert need > base
    need, base = _ray_scratch(7_201_012, 10_000_000)      # more than 8 M rays: the tables would take > 5 GB
    assert need == base
    need, base = _ray_scratch(30_000, 1_000_000)          # 16 levels, a big batch: too few subtrees to fill the chip
    assert need == base
    need, base = _ray_scratch(3_200, 1_000)               # ... a small batch on a small tree: binned again
    assert need > base
    need, base = _ray_scratch(3_200, 100_000)
    assert need == base
    need, base = _ray_scratch(250_000, 100_000)           # a medium tree: binned unless the rays outnumber its leaves 2 : 1
    assert need > base
    need, base = _ray_scratch(250_000, 1_000_000)
    assert need == base
    need, base = _ray_scratch(2_000, 500)                 # 12 levels: subtrees would hold fewer than 64 leaves
    assert need == base
    f64 = abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F64)
    need, base = _ray_scratch(7_201_012, 1_000_000, f64)  # Float64 throughout: binned as well (256-leaf subtrees)
    assert need > base
    mixed = abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F32)
    need, base = _ray_scratch(7_201_012, 1_000_000, mixed)  # two float types: ray traversal refuses those trees anyway
    assert need == base
    try:
        lib.set_tuning("rays_binned", 0)
        need, base = _ray_scratch(7_201_012, 1_000_000)
        assert need == base
        lib.set_tuning("rays_binned", 2)                  # wherever the tree allows it: a 1,000-leaf tree too
        need, base = _ray_scratch(1000, 5000)
        assert need > base
        lib.set_tuning("rays_items_per_ray", 4)
        small, _ = _ray_scratch(7_201_012, 1_000_000)
        assert 4 * 40 * 10**6 <= small < 16 * 40 * 10**6
    finally:
        lib.set_tuning("rays_binned", 1)
        lib.set_tuning("rays_items_per_ray", 0)


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


# ---------------------------------------------------------------------------------------------
# the Julia binding (source only: no Julia here) must agree with the ctypes binding the GPU tests run on
# ---------------------------------------------------------------------------------------------
def _julia_ext():
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return open(os.path.join(here, "implicitbvh.jl_amd", "julia", "ImplicitBVHlibibvhExt.jl")).read()


def test_julia_ccall_signatures_match_the_ctypes_binding():
    """Every `ccall((:ibvh_x, libibvh), Cint, (argtypes...), ...)` of the Julia extension has the argument kinds
    lib.SIGNATURES declares for ibvh_x (same count, same order, same width), so the two bindings cannot drift."""
    import ctypes as C
    import re
    from implicitbvh_amd import abi, lib
    src = _julia_ext()
    calls = re.findall(r"ccall\(\(:(\w+), libibvh\), (\w+),\s*\(([^)]*)\)", src)
    assert len(calls) >= 20
    jl = {"Ptr{Cvoid}": C.c_void_p, "Int64": C.c_int64, "Int32": C.c_int32, "Csize_t": C.c_size_t, "Float64": C.c_double,
          "Ref{Int64}": C.POINTER(C.c_int64), "Ref{Int32}": C.POINTER(C.c_int32), "Ref{Csize_t}": C.POINTER(C.c_size_t),
          "Ref{IbvhTypes}": C.POINTER(abi.Types), "Ref{IbvhTree}": C.POINTER(abi.Tree), "Ref{IbvhBvh}": C.POINTER(abi.Bvh),
          "Ref{IbvhBuildDesc}": C.POINTER(abi.BuildDesc), "Ref{IbvhBfsResult}": C.POINTER(abi.BfsResult),
          "Ref{IbvhComm}": C.POINTER(abi.Comm), "Ref{IbvhDistPlan}": C.POINTER(abi.DistPlan),
          "Ref{IbvhDistCrossPlan}": C.POINTER(abi.DistCrossPlan)}
    seen = set()
    for name, ret, args in calls:
        assert ret == ("Int32" if name == "ibvh_abi_version" else "Cint"), name
        got = [jl[a.strip()] for a in args.split(",") if a.strip()]
        assert got == lib.SIGNATURES[name], (name, args)
        seen.add(name)
    # the whole hot-path surface is bound: build + 3 LVT shapes x (count, write, enqueue) + 3 BFS shapes
    want = {"ibvh_build", "ibvh_build_scratch_bytes", "ibvh_lvt_scratch_bytes", "ibvh_lvt_total", "ibvh_bfs_counters_bytes",
            "ibvh_abi_version"}
    for shape in ("", "_pair", "_rays"):
        want |= {f"ibvh_traverse{shape}_lvt_{k}" for k in ("count", "write", "enqueue")}
        want |= {f"ibvh_traverse{shape}_bfs", f"ibvh_bfs{shape}_initial_capacity"}
    # ... and the distributed build (round 4: the driver is reachable from Julia with nothing but ccall)
    want |= {"ibvh_comm_from_rccl", "ibvh_dist_scratch_bytes", "ibvh_dist_plan", "ibvh_dist_exchange"}
    # ... and the cross-shard contact completion (round 5)
    want |= {f"ibvh_dist_cross_{k}" for k in ("plan", "exchange", "count", "write")}
    assert want <= seen, want - seen


def test_julia_pod_structs_match_the_header_mirror():
    """Field names, order and widths of the POD descriptors in the Julia extension == abi.py's ctypes mirror of ibvh.h."""
    import ctypes as C
    import re
    from implicitbvh_amd import abi
    src = _julia_ext()
    width = {"Int32": 4, "Int64": 8, "Ptr{Cvoid}": 8, "NTuple{3, Float64}": 24, "IbvhTypes": C.sizeof(abi.Types), "IbvhTree": C.sizeof(abi.Tree),
             "NTuple{6, Float64}": 48, "NTuple{256, UInt64}": 2048, "NTuple{256, Int64}": 2048, "NTuple{256, Int32}": 1024, "NTuple{24576, Float64}": 196608}
    for jname, ctype in (("IbvhTypes", abi.Types), ("IbvhTree", abi.Tree), ("IbvhBvh", abi.Bvh), ("IbvhBuildDesc", abi.BuildDesc),
                         ("IbvhBfsResult", abi.BfsResult), ("IbvhComm", abi.Comm), ("IbvhDistPlan", abi.DistPlan), ("IbvhDistCrossPlan", abi.DistCrossPlan)):
        body = re.search(r"struct " + jname + r"\n(.*?)\nend", src, re.S).group(1)
        fields = [(m.group(1), m.group(2).strip()) for m in re.finditer(r"(\w+)::([^;\n]+)", body)]
        assert [f for f, _ in fields] == [f for f, _ in ctype._fields_], jname
        assert [width[t] for _, t in fields] == [C.sizeof(t) for _, t in ctype._fields_], jname


def test_julia_extension_never_drops_narrow():
    """Round-1 finding: narrow_code(narrow) = 0 ignored user predicates.  Now unknown closures go to the generic method."""
    src = _julia_ext()
    assert "narrow_code(narrow) = Int32(0)" not in src
    assert src.count("isnothing(code)") >= 4 and src.count("invoke(ImplicitBVH.traverse") >= 4
    assert src.count("invoke(ImplicitBVH.traverse_rays") == 2


def _julia_calls(src, name):
    """argument text of every call `name(...)` in the extension (balanced parentheses, comments stripped)"""
    import re
    code = "\n".join(line.split("#")[0] for line in src.splitlines())
    out = []
    for m in re.finditer(r"(?<![\w.!])" + re.escape(name) + r"\(", code):
        depth, i = 1, m.end()
        while depth and i < len(code):
            depth += {"(": 1, ")": -1}.get(code[i], 0)
            i += 1
        out.append(code[m.end():i - 1])
    return out


def test_julia_constructor_calls_use_the_reference_signatures():
    """dist_BVH called constructors that do not exist.  Julia is not in the image, so the check is static:
    every `BVHOptions(...)` the extension builds passes only the reference's keywords (utils.jl:73-87: index, morton,
    num_threads, min_*_per_thread, block_size — there is no `index_exemplar` keyword, that is the FIELD), and every
    `DefaultMortonAlgorithm(...)` it builds is the one-positional keyword method (morton/default.jl:30-40: exemplar or type,
    then compute_extrema / mins / maxs by keyword; the four-positional inner constructor wants an exemplar VALUE)."""
    import re
    src = _julia_ext()
    ref_kw = {"index", "morton", "num_threads", "min_mortons_per_thread", "min_sorts_per_thread", "min_boundings_per_thread",
              "min_traversals_per_thread", "block_size"}
    calls = [c for c in _julia_calls(src, "BVHOptions") if c.strip()]
    assert calls, "dist_BVH builds its options"
    for c in calls:
        kws = set(re.findall(r"(?:^|[,;(\s])([a-z_]+)\s*=(?!=)", c))
        assert kws and kws <= ref_kw, (c, kws - ref_kw)
    morton = [c for c in _julia_calls(src, "DefaultMortonAlgorithm") if c.strip()]
    assert morton
    for c in morton:
        head, _, tail = c.partition(";")
        assert "," not in head, f"one positional argument, the rest by keyword: {c}"
        assert set(re.findall(r"([a-z_]+)\s*=(?!=)", tail)) <= {"compute_extrema", "mins", "maxs"}, c


def test_julia_pair_methods_fall_back_for_mixed_types():
    """INTEGRATION.md §4 "Pair traversal of two BVHs of DIFFERENT leaf / node types": both pair methods of the extension (LVT and
    BFS) test `d1.types != d2.types` and hand such pairs to the reference's generic method with every keyword carried over."""
    import re
    src = _julia_ext()
    for alg in ("LVTTraversal", "BFSTraversal"):
        m = re.search(r"function ImplicitBVH\.traverse\(\s*bvh1::RocBVH\{I\}, bvh2::RocBVH, alg::" + alg + r";(.*?)\nend\n", src, re.S)
        assert m, alg
        body = m.group(1)
        assert "d1.types != d2.types" in body
        call = re.search(r"invoke\(ImplicitBVH\.traverse, Tuple\{BVH, BVH, " + alg + r"\}, bvh1, bvh2, alg;(.*?)\)\n", body, re.S).group(1)
        for kw in ("start_level1", "start_level2", "narrow", "cache", "options"):
            assert f"{kw}={kw}" in call, (alg, kw)


def test_sort_hint_rule_of_a_rebuild_chain():
    """api.sort_hint_rule: what a `cache=` rebuild asks the library for (extra partition levels, equalised cells) from the hint
    word the previous build left (include/ibvh.h, ibvh_build_desc.sort_levels / sort_equalize / skew_flag).  Pure host logic."""
    from implicitbvh_amd import abi, api
    rule = api.sort_hint_rule
    n = 1_000_000
    # a comfortable uniform cloud: nothing; a nearly full cell: the spare level; beyond 2^24 leaves: always the spare level
    assert rule(60 << 8, n, 0) == (0, 0, 0)
    assert rule(api.SPARE_OCCUPANCY << 8, n, 0) == (1, 0, 0)
    assert rule(60 << 8, api.SPARE_ALWAYS_FROM, 0) == (1, 0, 0)
    # the previous build needed k levels: k + 1 of them and equalised cells
    for k in (1, 2, 3):
        assert rule(k | 200 << 8, n, 0) == (min(k + 1, abi.MAX_SORT_LEVELS), 1, 0)
    assert rule(abi.MAX_SORT_LEVELS, n, 0)[0] == abi.MAX_SORT_LEVELS
    # an equalised build that fitted (no level needed) while the plain grid would not have: equalised again, no idle spare level
    # unless its fullest cell was about full
    assert rule(1 << 16 | 100 << 8, n, 0) == (0, 1, 0)
    assert rule(1 << 16 | api.EQ_SPARE_OCCUPANCY << 8, n, 0) == (1, 1, 0)
    # ... and the plain grid would fit again: back to it (its own spare rule applies to the occupancy byte)
    assert rule(50 << 8, n, 0) == (0, 0, 0)
    # equalising did not help (bit 17): the plain grid for EQ_HOLDOFF rebuilds, whatever they report, then equalised cells again
    lv, eq, hold = rule(2 | 1 << 16 | 1 << 17 | 255 << 8, n, 0)
    assert (lv, eq, hold) == (3, 0, api.EQ_HOLDOFF)
    asked = []
    for _ in range(api.EQ_HOLDOFF):
        lv, eq, hold = rule(1 | 255 << 8, n, hold)
        asked.append(eq)
    assert asked == [0] * (api.EQ_HOLDOFF - 1) + [1] and hold == 0
    # round 6: an equalised build (bit 18) whose estimate says "the plain grid fits" (bit 16 = 0): back to the plain grid ON PROBATION,
    # with a spare level as insurance ...
    ran = 1 << 18
    assert rule(ran | 66 << 8, n, 0) == (1, 0, -1)
    # ... which fits: an ordinary plain chain again
    assert rule(60 << 8, n, -1) == (0, 0, 0)
    # ... or was crowded after all (the level caught it): equalised cells, sticky for EQ_STICKY rebuilds whatever bit 16 says
    lv, eq, hold = rule(1 | 170 << 8, n, -1)
    assert (lv, eq, hold) == (2, 1, -2 - api.EQ_STICKY)
    seen = []
    for _ in range(api.EQ_STICKY + 1):
        lv, eq, hold = rule(ran | 66 << 8, n, hold)
        seen.append((lv, eq))
    assert seen == [(0, 1)] * (api.EQ_STICKY + 1) and hold == 0
    assert rule(ran | 66 << 8, n, hold) == (1, 0, -1)  # then the estimate is tried again
    # a sticky chain whose equalised builds report that equalising does not help (bit 17) goes to the plain hold-off instead
    assert rule(ran | 1 << 17 | 2 | 255 << 8, n, -10) == (3, 0, api.EQ_HOLDOFF)
    # switched off as a whole
    try:
        api.EQUALIZE = False
        assert rule(2 | 1 << 16, n, 0)[1] == 0
    finally:
        api.EQUALIZE = True


def test_evidence_stamp_and_staleness_see_every_compiled_file(tmp_path):
    """`bench.csrc_sha()` (the stamp on the committed counter profiles) and `__graft_entry__.build()`'s staleness check
    cover every file the Makefile's compile rule depends on — including the textually included kernel bodies (*.inc:
    the dominant kernel lives in one), the Makefile's flags and include/ibvh.h. Editing a comment in the .inc alone in
    a temp copy must change the hash and make the library stale."""
    import shutil
    import time
    import bench
    import __graft_entry__ as entry
    root = tmp_path / "copy"
    csrc = root / "implicitbvh.jl_amd" / "csrc"
    csrc.mkdir(parents=True)
    (root / "include").mkdir()
    for f in entry.kernel_sources():
        rel = os.path.relpath(f, ROOT)
        shutil.copy2(f, root / rel)
    names = {os.path.basename(f) for f in entry.kernel_sources(str(root))}
    assert {"ibvh_lvt_queue.inc", "Makefile", "ibvh.h", "ibvh_common.hpp", "ibvh_build.hip"} <= names
    # the Makefile's own rule names the same dependencies
    mk = (csrc / "Makefile").read_text()
    rule = [l for l in mk.splitlines() if l.startswith("%.o:")][0]
    for dep in ("$(wildcard *.hpp)", "$(wildcard *.inc)", "../../include/ibvh.h", "Makefile"):
        assert dep in rule, rule
    so = root / "implicitbvh.jl_amd" / "libibvh.so"
    so.write_bytes(b"built")
    now = time.time()
    os.utime(so, (now, now))
    for f in entry.kernel_sources(str(root)):
        os.utime(f, (now - 100, now - 100))
    assert not entry.library_is_stale(str(root))
    assert bench.csrc_sha(str(root)) == bench.csrc_sha()
    for name, where in (("ibvh_lvt_queue.inc", csrc), ("Makefile", csrc), ("ibvh.h", root / "include")):
        before = bench.csrc_sha(str(root))
        p = where / name
        old = p.read_bytes()
        p.write_bytes(old + (b"\n# edited\n" if name == "Makefile" else b"\n// edited\n"))
        os.utime(p, (now + 100, now + 100))
        assert bench.csrc_sha(str(root)) != before, name
        assert entry.library_is_stale(str(root)), name
        p.write_bytes(old)
        os.utime(p, (now - 100, now - 100))
        assert bench.csrc_sha(str(root)) == before and not entry.library_is_stale(str(root))


def test_header_lists_exactly_the_knobs_the_library_has():
    """include/ibvh.h names the tuning knobs; the list must be the product table (csrc/ibvh_core.hip, outside the
    IBVH_VARIANTS block), and each name must be accepted — a variant-only name is refused."""
    hdr = open(os.path.join(ROOT, "include", "ibvh.h")).read()
    doc = hdr[hdr.index("Development knobs"):hdr.index("ibvh_status ibvh_set_tuning")]
    listed = set(re.findall(r'"([a-z0-9_]+)"', doc))
    core = open(os.path.join(ROOT, "implicitbvh.jl_amd", "csrc", "ibvh_core.hip")).read()
    table = core[core.index("const Knob kKnobs[]"):core.index("inline int64_t ilog2_down")]
    product = set(re.findall(r'\{"([a-z0-9_]+)"', re.sub(r"#ifdef IBVH_VARIANTS.*?#endif", "", table, flags=re.S)))
    assert listed == product, (listed ^ product)
    L = lib.load()
    v = C.c_int32()
    for name in sorted(product):
        assert L.ibvh_get_tuning(name.encode(), C.byref(v)) == 0, name
    for name in ("lvt_dual", "rays_shadow", "nope"):
        assert L.ibvh_get_tuning(name.encode(), C.byref(v)) != 0


# ---- the Julia extension as a faithful boundary (round 6) --------------------------------------------------------------
def _julia_methods(src):
    """{(generic function, first 120 chars of the signature): body} of every `function ImplicitBVH.x(` in the extension."""
    import re
    out = {}
    for m in re.finditer(r"\nfunction ImplicitBVH\.(\w+)\((.*?)\n\) where \{[^}]*\}\n(.*?)\nend\n", src, re.S):
        out[(m.group(1), " ".join(m.group(2).split())[:120])] = m.group(3)
    return out


def test_julia_unsupported_types_reach_the_generic_methods():
    """The extension's methods are more specific than the reference's, so every input the library has no instantiation for
    must be handed to the reference's generic method, not raise: BSphere{Float16} leaves (runtests.jl:480,510-538), an index
    type other than Int32 / Int64 (utils.jl:54-71 takes any Integer), a user-defined volume type, a Morton algorithm other
    than the default.  Checked statically (no Julia here): (1) every type-code table ends in a catch-all returning -1 and
    `native_types` turns any -1 into `nothing`; (2) in every method, each value obtained from `native_types(` / `bvh_desc(`
    is tested with `isnothing(...)` in an `if` that returns `invoke(<generic method>)` BEFORE its first use."""
    import re
    src = _julia_ext()
    assert "ibvh_types(" not in src  # (the old constructor that raised MethodError)
    # (1) the code tables, emulated: specific entries + catch-all
    tables = {}
    for fn in ("kind", "fltcode", "idxcode", "morcode"):
        entries = re.findall(fn + r"\(::Type(\{[^}]*\}+)?\)\s*=\s*Int32\((-?\d+)\)", src)
        tables[fn] = {(e[0] or "").strip("{}"): int(e[1]) for e in entries}
        assert tables[fn].get("") == -1, f"{fn} has no catch-all method returning -1"

    def code(fn, t):
        tab = tables[fn]
        if fn == "kind":
            return tab["<:BSphere"] if t.startswith("BSphere") else tab["<:BBox"] if t.startswith("BBox") else tab[""]
        return tab.get(t, tab[""])

    def native(leaf, node, index, morton):
        flt = lambda v: re.search(r"\{(\w+)\}", v).group(1) if re.match(r"(BSphere|BBox)\{", v) else "Any"
        codes = (code("kind", leaf), code("fltcode", flt(leaf)), code("kind", node), code("fltcode", flt(node)),
                 code("idxcode", index), code("morcode", morton))
        return None if any(c < 0 for c in codes) else codes
    nt = re.search(r"function native_types\(.*?\nend\n", src, re.S).group(0)
    assert "any(c -> c < 0, codes) && return nothing" in nt and "volume_float(L)" in nt and "volume_float(N)" in nt
    assert native("BSphere{Float32}", "BBox{Float32}", "Int32", "UInt32") == (0, 0, 1, 0, 0, 1)
    assert native("BBox{Float64}", "BSphere{Float64}", "Int64", "UInt64") == (1, 1, 0, 1, 1, 2)
    for bad in (("BSphere{Float16}", "BBox{Float32}", "Int32", "UInt32"), ("BSphere{Float32}", "BBox{Float16}", "Int32", "UInt32"),
                ("BSphere{Float32}", "BBox{Float32}", "UInt32", "UInt32"), ("BSphere{Float32}", "BBox{Float32}", "Int16", "UInt32"),
                ("MyCapsule", "BBox{Float32}", "Int32", "UInt32"), ("BSphere{Float32}", "BBox{Float32}", "Int32", "UInt128")):
        assert native(*bad) is None, bad
    # every supported combination of the Python mirror has codes here too (and the same ones)
    for k in (0, 1):
        for f, fname in ((0, "Float32"), (1, "Float64")):
            assert native(("BSphere", "BBox")[k] + "{" + fname + "}", "BBox{Float32}", "Int32", "UInt32")[:2] == (k, f)
    bd = re.search(r"function bvh_desc\(bvh::BVH\{I.*?\nend\n", src, re.S).group(0)
    assert "isnothing(types) && return nothing" in bd and "bvh_desc(bvh::BVH) = nothing" in src
    # (2) dominance in every method
    methods = _julia_methods(src)
    assert len(methods) == 7, sorted(methods)  # BVH + 2 x (single, pair, rays)
    generic = {"BVH": "invoke(ImplicitBVH.BVH, Tuple{AbstractVector, Type}", "traverse": "invoke(ImplicitBVH.traverse, Tuple{BVH",
               "traverse_rays": "invoke(ImplicitBVH.traverse_rays, Tuple{BVH, AbstractMatrix, AbstractMatrix"}
    for (fn, sig), body in methods.items():
        lines = [l for l in body.split("\n") if l.strip() and not l.strip().startswith("#")]
        text = "\n".join(lines)
        got = re.findall(r"^\s*([\w, ]+?)\s*=\s*(?:options\.morton isa DefaultMortonAlgorithm \? )?((?:bvh_desc|native_types)\(.*)$", text, re.M)
        assert got, (fn, sig)
        variables = []
        for lhs, rhs in got:
            variables += [v.strip() for v in lhs.split(",")]
        assert len(variables) == (2 if "bvh2" in sig else 1), (fn, sig, variables)
        if fn == "BVH":
            assert ": nothing" in got[0][1]  # a non-default Morton algorithm -> nothing -> generic
        for v in variables:
            guard = re.search(r"^\s*if [^\n]*isnothing\(" + v + r"\)[^\n]*\n\s*return " + re.escape(generic[fn]), text, re.M)
            assert guard, (fn, sig, v)
            before = text[:guard.start()]
            # before the guard the value is only assigned, never dereferenced or passed on
            uses = re.sub(r"(bvh_desc|native_types)\([^\n]*", "", before)   # (minus the assignments themselves)
            assert not re.search(r"\b" + v + r"\.", uses) and not re.search(r"[(,]\s*" + v + r"\s*[,)]", uses), (fn, sig, v)
        # the fallback carries every keyword over
        call = text[text.index(generic[fn]):]
        call = call[:call.index(")\n")]
        kws = ("built_level", "cache", "options") if fn == "BVH" else \
            ("start_level1", "start_level2", "narrow", "cache", "options") if "bvh2" in sig else ("start_level", "narrow", "cache", "options")
        for kw in kws:
            assert f"{kw}={kw}" in call, (fn, sig, kw)
    # BVH: the library's own verdict on a combination of known codes is a fallback too, and comes before any allocation / check
    body = methods[[k for k in methods if k[0] == "BVH"][0]]
    head = body[:body.index("return invoke(")]
    assert "== IBVH_ERR_UNSUPPORTED" in head and "similar(" not in head and "throw(" not in head and "ImplicitTree" not in head


def _julia_to_python(fn_src):
    """Transliterate one of the extension's two rule functions (written in a small subset on purpose) into Python."""
    import re
    lines = fn_src.strip("\n").split("\n")
    name, args = re.match(r"function (\w+)\((.*)\)$", lines[0]).groups()
    args = [a.split("::")[0].strip() for a in args.split(",")]
    out = [f"def {name}({', '.join(args)}):"]
    for l in lines[1:]:
        ind = len(l) - len(l.lstrip())
        t = l.strip()
        if t == "end" or not t or t.startswith("#"):
            continue
        t = t.replace("||", " or ").replace("&&", " and ")
        if t.startswith("if "):
            t = t + ":"
        elif t.startswith("elseif "):
            t = "elif " + t[len("elseif "):] + ":"
        elif t == "else":
            t = "else:"
        out.append(" " * ind + t)
    return "\n".join(out) + "\n"


def _julia_rules():
    import re
    src = _julia_ext()
    env = {"ifelse": lambda c, a, b: a if c else b, "cld": lambda a, b: -(-a // b), "min": min, "max": max,
           "nextpow": lambda base, x: 1 << max(0, (int(x) - 1).bit_length())}
    consts = dict(re.findall(r"^const ([A-Z_]+) = (\d+)\b", src, re.M))
    env.update({k: int(v) for k, v in consts.items()})
    for fn in ("sort_hint_rule", "cache_slots_rule"):
        code = re.search(r"^function " + fn + r"\(.*?^end$", src, re.S | re.M).group(0)
        exec(_julia_to_python(code), env)
    return env


def test_julia_build_policy_equals_the_python_mirror():
    """The host policies every BENCH number was measured with (api.sort_hint_rule: extra partition levels, equalised cells,
    the spare level, the hold-off; api._cache_slots: adaptive contact-cache slots) exist in the Julia extension and compute
    the same values: constants equal, the two rule functions transliterated and evaluated on a table of hint words /
    buffer sizes, and the build descriptor takes sort_levels / sort_equalize / skew_flag from them."""
    import itertools
    import re
    from implicitbvh_amd import abi, api
    env = _julia_rules()
    for name, want in (("COLD_SORT_LEVELS", api.COLD_SORT_LEVELS), ("SPARE_OCCUPANCY", api.SPARE_OCCUPANCY),
                       ("EQ_SPARE_OCCUPANCY", api.EQ_SPARE_OCCUPANCY), ("EQ_HOLDOFF", api.EQ_HOLDOFF), ("EQ_STICKY", api.EQ_STICKY),
                       ("SPARE_ALWAYS_FROM", api.SPARE_ALWAYS_FROM), ("EQUALIZE", int(api.EQUALIZE)),
                       ("MAX_SORT_LEVELS", abi.MAX_SORT_LEVELS), ("LVT_CACHE_SLOTS", api.LVT_CACHE_SLOTS),
                       ("RAY_CACHE_SLOTS", api.RAY_CACHE_SLOTS), ("HINT_WORDS", api._HostWords.HINT_SLOTS)):
        assert env[name] == want, name
    # every field of the hint word x sizes either side of SPARE_ALWAYS_FROM x hold-off states
    words = [u | o << 8 | e << 16 | h << 17 | r << 18 for u, o, e, h, r in
             itertools.product((0, 1, 2, 3, 4, 7), (0, 50, 95, 96, 119, 120, 128, 255), (0, 1), (0, 1), (0, 1))]
    for w, n, hold in itertools.product(words, (1, 250_000, 1_000_000, 10_000_000, (1 << 24) - 1, 1 << 24, 100_000_000),
                                        (0, 1, 2, api.EQ_HOLDOFF, -1, -2, -3, -2 - api.EQ_STICKY)):
        assert tuple(env["sort_hint_rule"](w, n, hold)) == tuple(api.sort_hint_rule(w, n, hold)), (hex(w), n, hold)
    # a chain, step by step, as the bench's rebuild loops drive it (the word each build leaves -> what the next one asks for)
    for chain in ([60 << 8] * 6,                                                  # uniform cloud
                  [2 | 255 << 8, 1 | 1 << 16 | 130 << 8, 1 << 16 | 100 << 8, 1 << 16 | 100 << 8],   # surface mesh
                  [3 | 1 << 16 | 1 << 17 | 255 << 8] + [1 | 255 << 8] * 40,       # runs of equal keys: hold-off and back
                  [1 | 170 << 8] + [1 << 18 | 66 << 8, 1 | 170 << 8] + [1 << 18 | 66 << 8] * 40):   # a mesh at the edge: probation, sticky
        hj = hp = 0
        for w in chain:
            lj, ej, hj = env["sort_hint_rule"](w, 1_000_000, hj)
            lp, ep, hp = api.sort_hint_rule(w, 1_000_000, hp)
            assert (lj, ej, hj) == (lp, ep, hp)

    class _Trav(api.BVHTraversal):
        def __init__(self, cap):
            self.cap = cap

        def _capacity(self):
            return self.cap
    for cap, n, default in itertools.product((0, 1, 7, 8, 9, 1000, 1_763_600, 7_000_000, 82_000_000, 10**9),
                                             (1, 1000, 1_000_000, 7_200_000), (api.LVT_CACHE_SLOTS, api.RAY_CACHE_SLOTS)):
        assert env["cache_slots_rule"](cap, n, default) == api._cache_slots(_Trav(cap), n, default), (cap, n, default)
    assert env["cache_slots_rule"](0, 1000, 8) == api._cache_slots(None, 1000, 8) == 8
    # the descriptor: its last three fields come from build_policy(cache, n), a cold build is (COLD_SORT_LEVELS, 0, own word)
    src = _julia_ext()
    body = _julia_methods(src)[[k for k in _julia_methods(src) if k[0] == "BVH"][0]]
    assert re.search(r"sort_levels, sort_equalize, skew_flag, chain = build_policy\(cache, numbv\)", body)
    desc = re.search(r"desc = IbvhBuildDesc\((.*?)\)\n\s*check\(c_build\(", body, re.S).group(1)
    assert desc.rstrip().endswith("sort_levels, sort_equalize, skew_flag")
    assert "chains[nodes] = chain" in body
    pol = re.search(r"^function build_policy\(cache, n\).*?^end$", src, re.S | re.M).group(0)
    assert "return (Int32(COLD_SORT_LEVELS), Int32(0), Ptr{Cvoid}(hint_ptr(st)), st)" in pol
    assert "sort_hint_rule(unsafe_load(hint_ptr(st), :monotonic), Int64(n), st.holdoff)" in pol
    # the traversal methods size the contact cache with the rule, not with the constant
    for k, b in _julia_methods(src).items():
        if "lvt_two_pass(" in b:
            assert re.search(r"lvt_two_pass\(I, [\w.]+, \w+, \w+\.types, cache_slots\(cache, \w+, (LVT|RAY)_CACHE_SLOTS\), cache,", b), k
