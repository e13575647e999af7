"""GPU parity tests: the HIP library (through its C ABI and the Python mirror) against the CPU oracle on
the same inputs.  Bit-exact for Morton codes, sorted order, node volumes (they are min/max/sqrt of
correctly-rounded ops, so exact equality is the bar, stricter than the north star's 1 ulp) and LVT
contact lists INCLUDING order; sorted-set equality for BFS (the reference's own GPU criterion,
test/gputests.jl:71-78)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi, lib  # noqa: E402

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))

NP_F = {abi.F32: np.float32, abi.F64: np.float64}
TOKENS = {abi.BSPHERE: ibvh.BSphere, abi.BBOX: ibvh.BBox}


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def random_volumes(rng, n, kind, flt, scale=6.0, size=1.0):
    f = NP_F[flt]
    c = (scale * rng.random((n, 3))).astype(f)
    if kind == abi.BSPHERE:
        r = (size * (0.1 + 0.9 * rng.random((n, 1)))).astype(f)
        return np.concatenate([c, r], axis=1)
    h = (size * (0.1 + 0.9 * rng.random((n, 3)))).astype(f)
    return np.concatenate([c - h, c + h], axis=1)


def make_options(types):
    return ibvh.BVHOptions(index=abi.INDEX_DTYPES[types.index_type],
                           morton=ibvh.DefaultMortonAlgorithm(abi.MORTON_DTYPES[types.morton_type]))


def build_both(vols, types, built_level=1, indices=None):
    o = orc.build(vols, types, built_level=built_level, indices=indices)
    node_type = TOKENS[types.node_kind](torch.float32 if types.node_float == abi.F32 else torch.float64)
    opts = make_options(types)
    if indices is None:
        g = ibvh.BVH(cuda(vols.astype(NP_F[types.leaf_float])), node_type, built_level=built_level, options=opts)
    else:
        bv = ibvh.BoundingVolumes.wrap(cuda(vols.astype(NP_F[types.leaf_float])), np.asarray(indices), opts)
        g = ibvh.BVH(bv, node_type, built_level=built_level, options=opts)
    return o, g


def assert_bvh_equal(o, g, built_level=1):
    gl = g.leaves.to_numpy()
    assert gl["morton"].tolist() == o.leaves["morton"].tolist()
    assert gl["index"].tolist() == o.leaves["index"].tolist()
    assert gl["volume"].tobytes() == o.leaves["volume"].tobytes()
    assert g.skips.cpu().numpy().tolist() == o.skips.tolist()
    assert g.extrema.cpu().numpy().tobytes() == o.extrema.tobytes()
    gn = g.nodes.cpu().numpy()
    on = o.nodes.view(gn.dtype).reshape(gn.shape) if len(o.nodes) else gn
    if len(o.nodes):
        lo = orc.memory_index(o.tree, 2 ** (min(built_level, o.tree.levels - 1) - 1)) - 1 if o.tree.levels > 1 else 0
        assert gn[lo:].tobytes() == on[lo:].tobytes()


ALL_COMBOS = [
    (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), (abi.BSPHERE, abi.F32, abi.BSPHERE, abi.F32),
    (abi.BBOX, abi.F32, abi.BBOX, abi.F32), (abi.BSPHERE, abi.F64, abi.BBOX, abi.F32),
    (abi.BSPHERE, abi.F64, abi.BBOX, abi.F64), (abi.BSPHERE, abi.F64, abi.BSPHERE, abi.F64),
    (abi.BSPHERE, abi.F64, abi.BSPHERE, abi.F32), (abi.BBOX, abi.F64, abi.BBOX, abi.F64),
    (abi.BBOX, abi.F64, abi.BBOX, abi.F32),
    # node float types WIDER than the leaves' (build.jl:198-205 builds any node_type; merge.jl computes in the promoted type)
    (abi.BSPHERE, abi.F32, abi.BBOX, abi.F64), (abi.BSPHERE, abi.F32, abi.BSPHERE, abi.F64), (abi.BBOX, abi.F32, abi.BBOX, abi.F64),
]


# ---------------------------------------------------------------------------------------------
# build
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("combo", ALL_COMBOS, ids=str)
@pytest.mark.parametrize("im", [(abi.I32, abi.U32), (abi.I64, abi.U64), (abi.I32, abi.U16)], ids=str)
def test_build_bit_exact_all_type_combinations(combo, im):
    rng = np.random.default_rng(hash((combo, im)) % 2**32)
    types = abi.make_types(*combo, *im)
    for n in (1, 2, 3, 5, 11, 64, 257, 1000, 4099):
        vols = random_volumes(rng, n, combo[0], combo[1])
        o, g = build_both(vols, types)
        assert_bvh_equal(o, g)


def test_morton_codes_equal_for_n_1_to_200():
    """test/gputests.jl:34-48."""
    rng = np.random.default_rng(0)
    types = abi.make_types()
    for n in range(1, 201):
        vols = random_volumes(rng, n, abi.BSPHERE, abi.F32, scale=1000.0)
        o, g = build_both(vols, types)
        assert g.leaves.to_numpy()["morton"].tolist() == o.leaves["morton"].tolist()


@pytest.mark.parametrize("n", [100003, 1 << 20, (1 << 22) + 12345])
def test_build_bit_exact_large(n):
    sph = orc.generate_spheres_f32(n, 42, r0=0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3))
    g_in = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3))
    assert g_in.cpu().numpy().tobytes() == sph.tobytes()  # device generator == host generator
    o, g = build_both(sph, abi.make_types())
    assert_bvh_equal(o, g)


def test_build_duplicate_codes_are_stable():
    """Many equal Morton codes: ties keep input order (stable sort, the oracle's documented choice)."""
    rng = np.random.default_rng(3)
    base = random_volumes(rng, 37, abi.BSPHERE, abi.F32)
    vols = np.repeat(base, 300, axis=0)  # 11100 leaves, 37 distinct centres
    rng.shuffle(vols)
    types = abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32, abi.I32, abi.U16)
    o, g = build_both(vols, types)
    assert_bvh_equal(o, g)
    o, g = build_both(vols, abi.make_types())
    assert_bvh_equal(o, g)


def test_build_prewrapped_keeps_user_indices_and_sorts_in_place():
    k = G["morton_kat"]
    vols = np.asarray(k["spheres"], np.float32)
    o, g = build_both(vols, abi.make_types(), indices=k["indices"])
    assert_bvh_equal(o, g)
    l1 = g.leaves.to_numpy()[0]
    assert int(l1["morton"]) == 0x06186186 and int(l1["index"]) == 3  # build.jl:136-152
    rng = np.random.default_rng(5)
    vols = random_volumes(rng, 5000, abi.BSPHERE, abi.F32)
    idx = rng.permutation(5000) + 100
    o, g = build_both(vols, abi.make_types(), indices=idx)
    assert_bvh_equal(o, g)


def test_build_levels_and_cache_reuse():
    rng = np.random.default_rng(9)
    vols = random_volumes(rng, 3000, abi.BSPHERE, abi.F32)
    types = abi.make_types()
    for bl in (1, 2, 5, orc.tree_shape(3000).levels - 1, orc.tree_shape(3000).levels):
        o, g = build_both(vols, types, built_level=bl)
        assert_bvh_equal(o, g, built_level=bl)
    g1 = ibvh.BVH(cuda(vols), ibvh.BBox(torch.float32))
    g2 = ibvh.BVH(cuda(vols), ibvh.BBox(torch.float32), cache=g1)
    assert g2.nodes.data_ptr() == g1.nodes.data_ptr() and g2.skips.data_ptr() == g1.skips.data_ptr()
    assert_bvh_equal(orc.build(vols, types), g2)
    g3 = ibvh.BVH(cuda(vols), ibvh.BBox(torch.float32), built_level=0.5)
    assert g3.built_level == orc.compute_build_level(orc.tree_shape(3000), 0.5)
    with pytest.raises(ValueError):
        ibvh.BVH(cuda(vols), ibvh.BBox(torch.float64), cache=g1)  # runtests.jl:915
    with pytest.raises(abi.DomainError):
        ibvh.BVH(torch.zeros((0, 4), device="cuda"))
    with pytest.raises(ValueError):
        ibvh.BVH(cuda(vols), built_level=99)


def test_fixed_extrema_option():
    rng = np.random.default_rng(10)
    vols = random_volumes(rng, 2000, abi.BSPHERE, abi.F32, scale=1.0, size=0.01)
    mins, maxs = (-0.5, -0.5, -0.5), (1.5, 1.5, 1.5)
    o = orc.build(vols, abi.make_types(), compute_extrema=False, mins=mins, maxs=maxs)
    opts = ibvh.BVHOptions(morton=ibvh.DefaultMortonAlgorithm(np.uint32, compute_extrema=False, mins=mins, maxs=maxs))
    g = ibvh.BVH(cuda(vols), options=opts)
    assert_bvh_equal(o, g)


# ---------------------------------------------------------------------------------------------
# stand-alone pieces through the raw C ABI
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key_bytes,key_bits", [(4, 30), (4, 15), (4, 32), (8, 63), (8, 40)])
@pytest.mark.parametrize("n", [1, 63, 2048, 2049, 100000, (1 << 22) + 77])
def test_sort_pairs_matches_stable_sort(key_bytes, key_bits, n):
    rng = np.random.default_rng(n + key_bits)
    kdt = np.uint32 if key_bytes == 4 else np.uint64
    # few distinct values in the low digit to stress stability
    keys = (rng.integers(0, 1 << min(key_bits, 62), n, dtype=np.uint64) & np.uint64(~np.uint64(0xF0))).astype(kdt)
    vals = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    ek, ev = orc.sort_pairs(keys, vals)
    k, v = cuda(keys.view(np.int32 if key_bytes == 4 else np.int64)), cuda(vals.view(np.int32))
    k2, v2 = torch.empty_like(k), torch.empty_like(v)
    need = C.c_size_t()
    lib.call("ibvh_sort_scratch_bytes", key_bytes, n, C.byref(need))
    scratch = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    in_alt = C.c_int32()
    lib.call("ibvh_sort_pairs", key_bytes, key_bits, n, k.data_ptr(), v.data_ptr(), k2.data_ptr(), v2.data_ptr(),
             C.byref(in_alt), scratch.data_ptr(), need.value, None)
    torch.cuda.synchronize()
    rk, rv = (k2, v2) if in_alt.value else (k, v)
    assert rk.cpu().numpy().view(kdt).tolist() == ek.tolist()
    assert rv.cpu().numpy().view(np.uint32).tolist() == ev.tolist()


@pytest.mark.parametrize("key_bytes,key_bits", [(4, 30), (8, 63)])
def test_sort_pairs_skewed_keys_take_the_oversized_bucket_path(key_bytes, key_bits):
    """Keys concentrated on a few values of the partition digit: buckets far larger than a workgroup's LDS
    capacity are sorted by the tiled in-bucket LSD; still a stable sort."""
    rng = np.random.default_rng(key_bits)
    kdt = np.uint32 if key_bytes == 4 else np.uint64
    for n in (50_000, 700_001):
        top = rng.choice(np.array([0, 1, 5, (1 << 11) - 1], dtype=np.uint64), n, p=[0.55, 0.3, 0.1, 0.05])
        low = rng.integers(0, 1 << 12, n, dtype=np.uint64) << np.uint64(3)  # few distinct low values too
        keys = ((top << np.uint64(key_bits - 11)) | low).astype(kdt)
        vals = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        ek, ev = orc.sort_pairs(keys, vals)
        k, v = cuda(keys.view(np.int32 if key_bytes == 4 else np.int64)), cuda(vals.view(np.int32))
        k2, v2 = torch.empty_like(k), torch.empty_like(v)
        need = C.c_size_t()
        lib.call("ibvh_sort_scratch_bytes", key_bytes, n, C.byref(need))
        scratch = torch.empty(need.value, dtype=torch.uint8, device="cuda")
        in_alt = C.c_int32()
        lib.call("ibvh_sort_pairs", key_bytes, key_bits, n, k.data_ptr(), v.data_ptr(), k2.data_ptr(), v2.data_ptr(),
                 C.byref(in_alt), scratch.data_ptr(), need.value, None)
        torch.cuda.synchronize()
        rk, rv = (k2, v2) if in_alt.value else (k, v)
        assert np.array_equal(rk.cpu().numpy().view(kdt), ek)
        assert np.array_equal(rv.cpu().numpy().view(np.uint32), ev)


def test_build_clustered_cloud_bit_exact():
    """Leaves piled into a few tight clusters: very uneven partition buckets in the hybrid sort."""
    rng = np.random.default_rng(99)
    centres = rng.random((6, 3)) * 100
    which = rng.choice(6, 300_000, p=[0.5, 0.3, 0.1, 0.05, 0.03, 0.02])
    c = centres[which] + rng.normal(0, 0.01, (300_000, 3))
    vols = np.concatenate([c, 0.001 + 0.002 * rng.random((300_000, 1))], axis=1).astype(np.float32)
    for mt in (abi.U32, abi.U64, abi.U16):
        o, g = build_both(vols, abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32, abi.I32, mt))
        assert_bvh_equal(o, g)


@pytest.mark.parametrize("shape", ["uniform", "clustered", "gaussian", "nested", "duplicates", "presorted", "reversed"])
def test_build_every_sort_route_at_every_partition_depth(shape):
    """The MSD build (n >= 4096): one partition level, up to four more for crowded cells (each on the key bits that
    vary inside the cell, straight into the output when at most 8 are left), and the tiled slow finish of whatever
    is still crowded after the last level that was launched — reached with every `sort_levels` from 0 to the maximum;
    the skew word tells the next build of the chain how many levels to launch."""
    rng = np.random.default_rng(len(shape) * 1000 + ord(shape[0]))
    n = 250_000
    if shape == "clustered":      # a few very tight clusters: most leaves share a handful of Morton codes
        centres = rng.random((3, 3)) * 50
        c = centres[rng.choice(3, n, p=[0.7, 0.2, 0.1])] + rng.normal(0, 1e-3, (n, 3))
    elif shape == "gaussian":     # clusters a few dozen grid cells wide: crowded cells AND crowded sub-cells
        centres = rng.random((8, 3))
        c = centres[rng.integers(0, 8, n)] + rng.normal(0, 0.004, (n, 3))
    elif shape == "nested":       # clusters inside clusters, down to exact duplicates
        sigma = rng.choice([0.05, 2e-3, 1e-4, 0.0], n, p=[0.2, 0.3, 0.3, 0.2])[:, None]
        c = np.array([[0.3, 0.6, 0.2]]) + sigma * rng.normal(0, 1, (n, 3))
        c[:100] = rng.random((100, 3))  # (a sparse background fixes the extent)
    elif shape == "duplicates":
        c = np.repeat(rng.random((37, 3)), n // 37 + 1, axis=0)[:n]
        rng.shuffle(c)
    else:
        c = rng.random((n, 3))
    vols = np.concatenate([c, 1e-4 + 1e-4 * rng.random((n, 1))], axis=1).astype(np.float32)
    if shape in ("presorted", "reversed"):
        order = orc.build(vols, abi.make_types()).leaves["index"].astype(np.int64) - 1
        vols = vols[order if shape == "presorted" else order[::-1]]
    expect_skew = shape in ("clustered", "gaussian", "nested", "duplicates")
    for combo, mt in [((abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.U32),
                      ((abi.BSPHERE, abi.F64, abi.BBOX, abi.F64), abi.U64)]:
        types = abi.make_types(*combo, abi.I32, mt)
        o, g = build_both(vols, types)                 # cold: COLD_SORT_LEVELS extra levels launched
        assert_bvh_equal(o, g)
        torch.cuda.synchronize()
        assert bool(int(g._skew[0])) == expect_skew
        node_type = TOKENS[types.node_kind](torch.float32 if types.node_float == abi.F32 else torch.float64)
        dev = cuda(vols.astype(NP_F[types.leaf_float]))
        # what "the previous build" left in the hint word (levels | fullest cell << 8): launches 0 extra levels (a comfortable
        # uniform cloud before), 1 (a nearly full cell before: the spare level), 2, 3, 4
        # (the PLAIN grid's levels and hint are what is pinned here: a chain whose hint shows skew would otherwise switch to
        # equalised cells — round 5, tests/test_gpu_sort_equalize.py and test_gpu_sort_stress.py — which need no level for most of
        # these inputs)
        from implicitbvh_amd import api
        try:
            api.EQUALIZE = False
            for pretend in (0, 127 << 8, 1, 2, 3):
                g._skew[0] = pretend
                g = ibvh.BVH(dev, node_type, options=make_options(types), cache=g)
                assert_bvh_equal(o, g)
                torch.cuda.synchronize()
                used = int(g._skew[0])
                assert (used > 0) == expect_skew and used <= abi.MAX_SORT_LEVELS
                if pretend == 0 and expect_skew:
                    assert used == 1                       # no extra level ran: all it can know is that one is needed
                if pretend == 127 << 8 and expect_skew:
                    assert used <= 2                       # one extra level ran: it can ask for at most one more
        finally:
            api.EQUALIZE = True
        # the same chain with equalised cells: the same bytes; only runs of equal keys longer than a finish workgroup sorts
        # still ask for a level
        for pretend in (1, 1 << 16, 3 | 1 << 16):
            g._skew[0] = pretend
            g = ibvh.BVH(dev, node_type, options=make_options(types), cache=g)
            assert_bvh_equal(o, g)
            torch.cuda.synchronize()
            assert int(g._fast[1].sort_equalize) == 1 if g._fast else True
            assert int(g._skew[0]) <= abi.MAX_SORT_LEVELS and (expect_skew or int(g._skew[0]) == 0)
        # no extra level at all (a raw C caller may pass sort_levels = 0): everything crowded takes the one-workgroup path
        from implicitbvh_amd import api
        saved = api.COLD_SORT_LEVELS
        try:
            api.COLD_SORT_LEVELS = 0
            g0 = ibvh.BVH(dev, node_type, options=make_options(types))
            assert_bvh_equal(o, g0)
            torch.cuda.synchronize()
            used = int(g0._skew[0])
            assert (used > 0) == expect_skew
            if expect_skew:
                assert used == 1                       # no extra level ran: all it can know is that one is needed
        finally:
            api.COLD_SORT_LEVELS = saved


def _kernels_of_one_cached_build(dev, cache):
    lib.call("ibvh_profile_enable", 1)
    g = ibvh.BVH(dev, cache=cache)
    torch.cuda.synchronize()
    cnt = C.c_int64()
    lib.call("ibvh_profile_count", C.byref(cnt))
    names = set()
    for i in range(cnt.value):
        name, ms = C.c_char_p(), C.c_float()
        lib.call("ibvh_profile_get", i, C.byref(name), C.byref(ms))
        names.add(name.value.decode())
    lib.call("ibvh_profile_enable", 0)
    return g, " ".join(sorted(names))


def test_spare_sort_level_follows_the_fullest_cell(monkeypatch):
    """The hint word's second byte is the fullest coarse cell in 1/128 of a finish workgroup's capacity; a cached build
    launches the spare extra level (four more kernels) only from api.SPARE_OCCUPANCY on, or when the chain used extra levels."""
    from implicitbvh_amd import api
    rng = np.random.default_rng(11)
    n = 300_000
    vols = np.concatenate([rng.random((n, 3)), 1e-4 * np.ones((n, 1))], axis=1).astype(np.float32)
    o = orc.build(vols, abi.make_types())
    dev = cuda(vols)
    g = ibvh.BVH(dev)
    torch.cuda.synchronize()
    assert int(g._skew[0]) == 0 and 32 <= g._skew.occupancy() < 128     # uniform: the average cell is 30 - 60 % full
    occ = g._skew.occupancy()
    monkeypatch.setattr(api, "SPARE_OCCUPANCY", occ + 1)                 # comfortably below: no extra level
    g, names = _kernels_of_one_cached_build(dev, g)
    assert "hist_level_kernel" not in names and "finish_kernel" in names
    assert_bvh_equal(o, g)
    monkeypatch.setattr(api, "SPARE_OCCUPANCY", occ)                     # at the threshold: the spare level runs
    g, names = _kernels_of_one_cached_build(dev, g)
    assert "hist_level_kernel" in names
    assert_bvh_equal(o, g)
    assert g._skew.occupancy() == occ
    # a clustered cloud through the same chain while the hint says "comfortable": no extra level, still exact; the hint it
    # leaves turns the levels on for the build after
    monkeypatch.setattr(api, "SPARE_OCCUPANCY", 128)
    c = rng.random((5, 3))[rng.integers(0, 5, n)] + rng.normal(0, 2e-4, (n, 3))
    clustered = np.concatenate([c, 1e-4 * np.ones((n, 1))], axis=1).astype(np.float32)
    oc = orc.build(clustered, abi.make_types())
    g, names = _kernels_of_one_cached_build(cuda(clustered), g)
    assert "hist_level_kernel" not in names
    assert_bvh_equal(oc, g)
    assert int(g._skew[0]) == 1 and g._skew.occupancy() == 255
    g, names = _kernels_of_one_cached_build(cuda(clustered), g)
    assert "hist_level_kernel" in names
    assert_bvh_equal(oc, g)


@pytest.mark.parametrize("levels", [-1, 99])
def test_build_accepts_any_sort_levels_value(levels, monkeypatch):
    """ibvh_build_desc.sort_levels outside 0 .. IBVH_MAX_SORT_LEVELS means "all of them" (include/ibvh.h)."""
    from implicitbvh_amd import api
    monkeypatch.setattr(api, "COLD_SORT_LEVELS", levels)
    rng = np.random.default_rng(5)
    c = rng.random((3, 3))[rng.integers(0, 3, 60_000)] + rng.normal(0, 1e-3, (60_000, 3))
    vols = np.concatenate([c, 1e-4 * np.ones((60_000, 1))], axis=1).astype(np.float32)
    o, g = build_both(vols, abi.make_types())
    assert_bvh_equal(o, g)
    torch.cuda.synchronize()
    assert 1 <= int(g._skew[0]) <= abi.MAX_SORT_LEVELS


def test_extrema_and_keys_entry_points():
    rng = np.random.default_rng(12)
    for kind, flt in ((abi.BSPHERE, abi.F32), (abi.BBOX, abi.F64)):
        types = abi.make_types(kind, flt, abi.BBOX, abi.F32, abi.I32, abi.U64)
        vols = random_volumes(rng, 12345, kind, flt, scale=-50.0)  # all-negative centres: floatmin max-init quirk
        sv = orc.as_volumes(vols, kind, flt)
        e = orc.extrema(types, sv, wrapped=False)
        assert (e[3:] == np.finfo(NP_F[flt]).tiny * 2).all() or (e[3:] > 0).all()
        dv = cuda(vols)
        ext = torch.empty(6, dtype=dv.dtype, device="cuda")
        scratch = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
        lib.call("ibvh_extrema", C.byref(types), dv.data_ptr(), 0, len(vols), 1, ext.data_ptr(), scratch.data_ptr(),
                 scratch.numel(), None)
        assert ext.cpu().numpy().tobytes() == e.tobytes()
        keys = torch.empty(len(vols), dtype=torch.int64, device="cuda")
        lib.call("ibvh_morton_keys", C.byref(types), dv.data_ptr(), 0, len(vols), ext.data_ptr(), keys.data_ptr(), None)
        assert keys.cpu().numpy().view(np.uint64).tolist() == orc.morton_keys(types, sv, False, e).tolist()


# ---------------------------------------------------------------------------------------------
# LVT: contact lists identical INCLUDING order
# ---------------------------------------------------------------------------------------------
def contacts_np(trav):
    return trav.contacts.cpu().numpy().astype(np.int64)


def oracle_pairs(c):
    return np.stack([c["a"], c["b"]], axis=1).astype(np.int64) if len(c) else np.zeros((0, 2), np.int64)


@pytest.mark.parametrize("combo", ALL_COMBOS, ids=str)
def test_lvt_self_identical_order_every_start_level(combo):
    rng = np.random.default_rng(21)
    types = abi.make_types(*combo)
    for n in (1, 2, 5, 12, 100, 1001):
        vols = random_volumes(rng, n, combo[0], combo[1])
        o, g = build_both(vols, types)
        for sl in range(1, o.tree.levels + 1):
            exp = oracle_pairs(orc.traverse_lvt(o, sl)[0])
            got = ibvh.traverse(g, ibvh.LVTTraversal(), start_level=sl)
            assert got.num_contacts == len(exp)
            assert (contacts_np(got) == exp).all(), (n, sl)
        brute = sorted(map(tuple, orc.brute_force_self(combo[0], combo[1], vols).tolist()))
        assert sorted(map(tuple, contacts_np(ibvh.traverse(g)).tolist())) == brute


@pytest.mark.parametrize("combo", [c for c in ALL_COMBOS if c[2] == abi.BBOX], ids=str)
def test_lvt_queue_kernel_medium_clouds_all_box_combinations(combo):
    """The BBox-node fast kernel at a size where every wave runs the full pipeline (frontier descent, both candidate
    loops, queue drains, dense cache) for every leaf / node float combination — Float32 nodes take the hand-scheduled
    step, Float64 nodes the compiler's — with both index types, the narrow menu, and a pair traversal with flip."""
    rng = np.random.default_rng(31)
    n = 40000
    vols = random_volumes(rng, n, combo[0], combo[1], scale=22.0)
    for it, mt in ((abi.I32, abi.U32), (abi.I64, abi.U64)):
        types = abi.make_types(*combo, it, mt)
        o, g = build_both(vols, types)
        exp = oracle_pairs(orc.traverse_lvt(o)[0])
        assert len(exp) > n // 2
        t = ibvh.traverse(g)
        assert (contacts_np(t) == exp).all()
        assert (contacts_np(ibvh.traverse(g, cache=t)) == exp).all()
        for nar in (ibvh.NARROW_MORTON_LT, ibvh.NARROW_INDEX_LT):
            assert (contacts_np(ibvh.traverse(g, narrow=nar)) == oracle_pairs(orc.traverse_lvt(o, narrow=nar)[0])).all()
    types = abi.make_types(*combo)
    other = random_volumes(rng, n // 3, combo[0], combo[1], scale=22.0)
    (o1, g1), (o2, g2) = build_both(vols, types), build_both(other, types)
    assert (contacts_np(ibvh.traverse(g1, g2)) == oracle_pairs(orc.traverse_pair_lvt(o1, o2)[0])).all()
    assert (contacts_np(ibvh.traverse(g2, g1, narrow=ibvh.NARROW_INDEX_LT)) ==
            oracle_pairs(orc.traverse_pair_lvt(o2, o1, narrow=abi.NARROW_INDEX_LT)[0])).all()


@pytest.mark.parametrize("flt", [abi.F32, abi.F64], ids=["f32", "f64"])
def test_extreme_value_inputs_bit_exact(flt):
    """Inputs at the edges of the float range: denormal-sized and 1e30-sized scenes (squares overflow to inf), zero and
    negative radii, all leaves identical, a large offset, all-negative coordinates (the floatmin max-init quirk).  Codes,
    order, extrema, every node and the contact list must still equal the oracle's, bit for bit."""
    f = NP_F[flt]
    rng = np.random.default_rng(5)
    types = abi.make_types(abi.BSPHERE, flt, abi.BBOX, flt)
    n = 3000
    base = np.concatenate([rng.random((n, 3)), 0.02 + 0.03 * rng.random((n, 1))], axis=1)
    tiny, huge = (1e-38, 1e30) if flt == abi.F32 else (1e-300, 1e250)
    zero_r, neg_r, offset, negative = base.copy(), base.copy(), base.copy(), base.copy()
    zero_r[:, 3] = 0
    neg_r[::3, 3] *= -1
    offset[:, :3] += 1e6
    negative[:, :3] -= 5
    for name, vols in (("denormal", base * tiny), ("huge", base * huge), ("zero radius", zero_r), ("negative radii", neg_r),
                       ("identical", np.tile(base[:1], (500, 1))), ("offset", offset), ("negative", negative)):
        o, g = build_both(vols.astype(f), types)
        assert_bvh_equal(o, g)
        exp = oracle_pairs(orc.traverse_lvt(o)[0])
        assert (contacts_np(ibvh.traverse(g)) == exp).all(), name
        # (BFS against the oracle's BFS, not against LVT: where squares underflow to zero the leaf test passes for any
        # pair that is reached, and the two traversals prune differently on the way down)
        eb, res = orc.traverse_bfs(o)
        bfs = ibvh.traverse(g, ibvh.BFSTraversal())
        assert sorted(map(tuple, contacts_np(bfs).tolist())) == sorted(map(tuple, oracle_pairs(eb).tolist())), name
        assert bfs.num_checks == res.num_checks, name


def test_readme_examples_on_gpu():
    e = G["readme_example"]
    for dt in (np.float32, np.float64):
        g = ibvh.BVH(cuda(np.asarray(e["spheres"], dt)))
        t = ibvh.traverse(g)
        assert contacts_np(t).tolist() == e["contacts"]
        t = ibvh.traverse(g, cache=t)
        assert contacts_np(t).tolist() == e["contacts"]
        assert sorted(contacts_np(ibvh.traverse(g, ibvh.BFSTraversal())).tolist()) == sorted(e["contacts"])
    p = G["pair_example"]
    b1, b2 = ibvh.BVH(cuda(np.asarray(p["spheres1"], np.float32))), ibvh.BVH(cuda(np.asarray(p["spheres2"], np.float32)))
    t = ibvh.traverse(b1, b2, start_level1=p["start_level1"], start_level2=p["start_level2"])
    assert contacts_np(t).tolist() == p["contacts"]
    r = G["ray_example"]
    g = ibvh.BVH(cuda(np.asarray(r["spheres"], np.float32)))
    pts = torch.tensor(r["points"], dtype=torch.float64).t()
    dirs = torch.tensor(r["directions"], dtype=torch.float64).t()
    t = ibvh.traverse_rays(g, pts, dirs)
    assert contacts_np(t).tolist() == r["contacts"]
    assert sorted(contacts_np(ibvh.traverse_rays(g, pts, dirs, ibvh.BFSTraversal())).tolist()) == sorted(r["contacts"])


def test_lvt_self_large_and_index_types():
    n = 200000
    r0 = 0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3)
    sph = orc.generate_spheres_f32(n, 7, r0=r0)
    for it, mt in ((abi.I32, abi.U32), (abi.I64, abi.U64)):
        types = abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32, it, mt)
        o, g = build_both(sph, types)
        exp = oracle_pairs(orc.traverse_lvt(o)[0])
        got = ibvh.traverse(g)
        assert (contacts_np(got) == exp).all()
        assert got.cache2.cpu().numpy().tolist() == orc.traverse_lvt(o)[1].tolist()  # inclusive prefix counts
        assert len(exp) > n  # ~1.8 contacts per leaf


def test_lvt_dense_inputs_cache_overflow_and_frontier_overflow():
    """Paths of the BBox fast kernel that sparse clouds never take: work items with more contacts than the
    contact cache holds (the writing pass walks again) and frontiers wider than the LDS buffer (the wave
    falls back to the exact joint walk).  Lists must still equal the oracle's, order included."""
    rng = np.random.default_rng(77)
    types = abi.make_types()
    # ~60 contacts per leaf
    vols = random_volumes(rng, 4000, abi.BSPHERE, abi.F32, scale=4.0, size=1.0)
    o, g = build_both(vols, types)
    exp = oracle_pairs(orc.traverse_lvt(o)[0])
    assert len(exp) > 20 * 4000
    assert (contacts_np(ibvh.traverse(g)) == exp).all()
    for sl in (3, o.tree.levels - 1, o.tree.levels):
        assert (contacts_np(ibvh.traverse(g, start_level=sl)) == oracle_pairs(orc.traverse_lvt(o, sl)[0])).all()
    # everything overlaps everything: every node of every level is hit
    vols = random_volumes(rng, 3000, abi.BSPHERE, abi.F32, scale=0.05, size=1.0)
    o, g = build_both(vols, types)
    exp = oracle_pairs(orc.traverse_lvt(o)[0])
    assert len(exp) == 3000 * 2999 // 2
    assert (contacts_np(ibvh.traverse(g)) == exp).all()
    # pair version, both orders (flip)
    a, b = random_volumes(rng, 1500, abi.BSPHERE, abi.F32, scale=2.0), random_volumes(rng, 900, abi.BSPHERE, abi.F32, scale=2.0)
    (o1, g1), (o2, g2) = build_both(a, types), build_both(b, types)
    assert (contacts_np(ibvh.traverse(g1, g2)) == oracle_pairs(orc.traverse_pair_lvt(o1, o2)[0])).all()
    assert (contacts_np(ibvh.traverse(g2, g1)) == oracle_pairs(orc.traverse_pair_lvt(o2, o1)[0])).all()
    # BSphere nodes (exact joint walk) on a larger cloud
    ts = abi.make_types(abi.BSPHERE, abi.F32, abi.BSPHERE, abi.F32)
    vols = random_volumes(rng, 30000, abi.BSPHERE, abi.F32, scale=20.0)
    o, g = build_both(vols, ts)
    assert (contacts_np(ibvh.traverse(g)) == oracle_pairs(orc.traverse_lvt(o)[0])).all()


def test_lvt_enqueue_without_host_sync_matches_the_two_call_protocol():
    """ibvh_traverse_lvt_enqueue (count + scan + guarded writing pass, no host read in between) through the C ABI:
    same counts, same contacts as _count/_write when the buffer is large enough; nothing written and the right
    total when it is not, after which _write completes the job."""
    rng = np.random.default_rng(5)
    types = abi.make_types()
    vols = random_volumes(rng, 20000, abi.BSPHERE, abi.F32, scale=14.0)
    o, g = build_both(vols, types)
    exp, exp_counts = orc.traverse_lvt(o)
    exp = oracle_pairs(exp)
    n = len(vols)
    s = g.struct()
    need = C.c_size_t()
    lib.call("ibvh_lvt_scratch_bytes", C.byref(types), n, 8, C.byref(need))
    scratch = torch.zeros(need.value, dtype=torch.uint8, device="cuda")
    counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    # large enough
    cap = len(exp) + 100
    contacts = torch.full((cap, 2), -7, dtype=torch.int32, device="cuda")
    lib.call("ibvh_traverse_lvt_enqueue", C.byref(s), 1, 0, counts.data_ptr(), contacts.data_ptr(), cap, None, None,
             scratch.data_ptr(), scratch.numel(), stream)  # total_dev = total_host = NULL: the total stays in the scratch header
    total = C.c_int64()
    lib.call("ibvh_lvt_total", scratch.data_ptr(), C.byref(total), stream)
    assert total.value == len(exp)
    assert counts.cpu().numpy().tolist() == exp_counts.tolist()
    got = contacts.cpu().numpy()
    assert (got[: len(exp)] == exp).all() and (got[len(exp):] == -7).all()
    # exactly enough, then one too few: the guarded pass must not touch the buffer
    for cap, written in ((len(exp), True), (len(exp) - 1, False)):
        contacts = torch.full((len(exp), 2), -7, dtype=torch.int32, device="cuda")
        tdev = torch.full((3,), -1, dtype=torch.int64, device="cuda")  # the caller's own total word (middle one)
        # ... and a word of mapped pinned host memory that the scan kernel fills as well (total_host): polled, no sync
        thost = torch.full((3,), -5, dtype=torch.int64).pin_memory()
        lib.call("ibvh_traverse_lvt_enqueue", C.byref(s), 1, 0, counts.data_ptr(), contacts.data_ptr(), cap,
                 tdev.data_ptr() + 8, thost.data_ptr() + 8, scratch.data_ptr(), scratch.numel(), stream)
        import time
        t_end = time.time() + 30
        view = thost.numpy()
        while int(view[1]) == -5 and time.time() < t_end:
            pass
        assert view.tolist() == [-5, len(exp), -5]
        lib.call("ibvh_lvt_total", tdev.data_ptr() + 8, C.byref(total), stream)
        assert tdev.cpu().tolist() == [-1, len(exp), -1]
        assert total.value == len(exp)
        got = contacts.cpu().numpy()
        assert (got == exp).all() if written else (got == -7).all()
    lib.call("ibvh_traverse_lvt_write", C.byref(s), 1, 0, counts.data_ptr(), contacts.data_ptr(), scratch.data_ptr(),
             scratch.numel(), stream)
    assert (contacts.cpu().numpy() == exp).all()
    # capacity 0: counting only
    lib.call("ibvh_traverse_lvt_enqueue", C.byref(s), 1, 0, counts.data_ptr(), None, 0, None, None, scratch.data_ptr(), scratch.numel(), stream)
    lib.call("ibvh_lvt_total", scratch.data_ptr(), C.byref(total), stream)
    assert total.value == len(exp)


def test_lvt_cache_reuse_is_lazy_and_grows_when_the_cached_buffer_is_too_small():
    """Python mirror: traverse(bvh; cache=previous) enqueues against the cached contact buffer and reads the
    count only when asked; a buffer that turns out too small is replaced at that point."""
    rng = np.random.default_rng(6)
    types = abi.make_types()
    small = random_volumes(rng, 3000, abi.BSPHERE, abi.F32, scale=30.0)   # few contacts
    big = random_volumes(rng, 30000, abi.BSPHERE, abi.F32, scale=14.0)    # many
    (os_, gs), (ob, gb) = build_both(small, types), build_both(big, types)
    exp_s, exp_b = oracle_pairs(orc.traverse_lvt(os_)[0]), oracle_pairs(orc.traverse_lvt(ob)[0])
    assert 0 < len(exp_s) < len(exp_b)
    t1 = ibvh.traverse(gs)
    assert (contacts_np(t1) == exp_s).all()
    t2 = ibvh.traverse(gb, cache=t1)            # cached buffer too small: resolved on first access
    assert t2._pending is not None
    assert t2.num_contacts == len(exp_b) and (contacts_np(t2) == exp_b).all()
    t3 = ibvh.traverse(gb, cache=t2)            # large enough now: written speculatively
    assert t3._pending is not None
    assert (contacts_np(t3) == exp_b).all() and t3.cache1.data_ptr() == t2.cache1.data_ptr()
    t4 = ibvh.traverse(gs, cache=t3)            # shrinking: same buffer, fewer contacts
    assert (contacts_np(t4) == exp_s).all()
    # an unread traversal whose buffers were handed on and whose cached buffer was too small cannot be completed
    t5 = ibvh.traverse(gs)
    t6 = ibvh.traverse(gb, cache=t5)
    t7 = ibvh.traverse(gb, cache=t6)
    with pytest.raises(RuntimeError):
        t6.num_contacts
    assert (contacts_np(t7) == exp_b).all()


def test_lvt_pending_count_survives_later_traversals_on_the_same_cache():
    """A traversal enqueued with `cache=` keeps its total in a ring of per-call device words, NOT in the scratch the
    next calls rewrite (round-1 advisor finding: header slots inside the scratch were overwritten by the tile sums of
    later calls).  Chains of more than 64 calls on one cache, a late read just before and just after the ring wraps,
    and — with more than 2 M items, where the tile sums alone exceed 4 KiB — a read one call late."""
    rng = np.random.default_rng(16)
    types = abi.make_types()
    small = random_volumes(rng, 4000, abi.BSPHERE, abi.F32, scale=20.0)
    other = random_volumes(rng, 5000, abi.BSPHERE, abi.F32, scale=20.0)
    (o1, g1), (o2, g2) = build_both(small, types), build_both(other, types)
    n1, n2 = len(orc.traverse_lvt(o1)[0]), len(orc.traverse_lvt(o2)[0])
    assert n1 != n2 and n1 > 0 and n2 > 0
    t = ibvh.traverse(g2)                       # sizes the cached buffers for the larger list
    kept = []
    for k in range(130):                        # two wraps of the 64-entry ring
        t = ibvh.traverse(g1 if k % 2 == 0 else g2, cache=t)
        assert t._pending is not None
        kept.append((k, t))
        if len(kept) == 60:                     # read 59 ... 1 calls late: all still valid
            for kk, tt in kept:
                assert tt._pending[0].item() == (n1 if kk % 2 == 0 else n2)
            kept = []
    stale = kept[0][1]                          # enqueued 70 - 1 calls ago by now? make it so
    for _ in range(70):
        t = ibvh.traverse(g2, cache=t)
    with pytest.raises(RuntimeError):
        stale._pending[0].item()                # recycled: refuses instead of returning another call's number
    assert t.num_contacts == n2
    # > 2 M items: the scan's tile sums reach beyond the first 4 KiB of the scratch
    nbig = 2_200_000
    big = ibvh.generate_spheres(nbig, 5, r0=0.5 * (3 * 8 / (4 * np.pi * nbig)) ** (1 / 3))
    gb = ibvh.BVH(big)
    ref = ibvh.traverse(gb)
    want = ref.num_contacts
    tb = ibvh.traverse(gb, cache=ref)
    for _ in range(66):                          # across the wrap of the ring
        prev = tb
        tb = ibvh.traverse(gb, cache=tb)
        assert prev._pending[0].item() == want   # read one call late


def test_lvt_pair_and_rays_cache_reuse_enqueue_paths():
    """pair and ray traversals with `cache=`: enqueued against the cached buffer (ibvh_traverse_pair_lvt_enqueue /
    ibvh_traverse_rays_lvt_enqueue), same lists as the oracle, both when the buffer fits and when it must grow."""
    rng = np.random.default_rng(8)
    types = abi.make_types()
    a, b = random_volumes(rng, 5000, abi.BSPHERE, abi.F32, scale=5.0), random_volumes(rng, 3000, abi.BSPHERE, abi.F32, scale=5.0)
    (o1, g1), (o2, g2) = build_both(a, types), build_both(b, types)
    exp12, exp21 = oracle_pairs(orc.traverse_pair_lvt(o1, o2)[0]), oracle_pairs(orc.traverse_pair_lvt(o2, o1)[0])
    t = ibvh.traverse(g1, g2)
    assert (contacts_np(t) == exp12).all()
    t = ibvh.traverse(g2, g1, cache=t)
    assert t._pending is not None and (contacts_np(t) == exp21).all()
    tiny = ibvh.traverse(g1, g2, cache=ibvh.traverse(ibvh.BVH(cuda(a[:40]), ibvh.BBox(torch.float32))))  # buffer too small
    assert (contacts_np(tiny) == exp12).all()
    pts = (5.0 * rng.random((3, 700))).astype(np.float32)
    dirs = rng.standard_normal((3, 700)).astype(np.float32)
    exp_r = oracle_pairs(orc.traverse_rays_lvt(o1, np.ascontiguousarray(pts.T), np.ascontiguousarray(dirs.T))[0])  # oracle: (N, 3)
    r = ibvh.traverse_rays(g1, cuda(pts), cuda(dirs))
    assert (contacts_np(r) == exp_r).all()
    r2 = ibvh.traverse_rays(g1, cuda(pts[:, :300]), cuda(dirs[:, :300]), cache=r)
    assert r2._pending is not None
    assert (contacts_np(r2) == oracle_pairs(orc.traverse_rays_lvt(o1, np.ascontiguousarray(pts[:, :300].T),
                                                                 np.ascontiguousarray(dirs[:, :300].T))[0])).all()


def test_lvt_pair_identical_order():
    rng = np.random.default_rng(22)
    types = abi.make_types()
    for n1, n2 in ((1, 1), (1, 50), (50, 1), (22, 190), (190, 22), (300, 300), (1000, 777)):
        a, b = random_volumes(rng, n1, abi.BSPHERE, abi.F32), random_volumes(rng, n2, abi.BSPHERE, abi.F32)
        o1, g1 = build_both(a, types)
        o2, g2 = build_both(b, types)
        for sl1 in sorted({1, o1.tree.levels // 2 + 1, o1.tree.levels}):
            for sl2 in sorted({1, o2.tree.levels // 2 + 1, o2.tree.levels}):
                exp = oracle_pairs(orc.traverse_pair_lvt(o1, o2, sl1, sl2)[0])
                got = ibvh.traverse(g1, g2, start_level1=sl1, start_level2=sl2)
                assert (contacts_np(got) == exp).all(), (n1, n2, sl1, sl2)
        brute = sorted(map(tuple, orc.brute_force_pair(abi.BSPHERE, abi.F32, a, b).tolist()))
        assert sorted(map(tuple, contacts_np(ibvh.traverse(g1, g2)).tolist())) == brute
    # the contract for BVHs of different leaf / node types: the status, never a wrong list
    _, gb = build_both(random_volumes(rng, 100, abi.BBOX, abi.F32), abi.make_types(abi.BBOX, abi.F32, abi.BBOX, abi.F32))
    s1, s2 = g1.struct(), gb.struct()
    counts = torch.zeros(1000, dtype=torch.int32, device="cuda")
    scratch = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    total = C.c_int64()
    rc = lib.load().ibvh_traverse_pair_lvt_count(C.byref(s1), C.byref(s2), 1, 1, 0, counts.data_ptr(), C.byref(total), scratch.data_ptr(),
                                                 scratch.numel(), None)
    assert rc == abi.ERR_UNSUPPORTED


@pytest.mark.parametrize("slots", [8, 0])
@pytest.mark.parametrize("idx", [abi.I32, abi.I64], ids=["i32", "i64"])
def test_lvt_pair_clouds_that_miss_or_barely_touch_each_others_root_box(slots, idx, monkeypatch):
    """The pair walk lets a wave leave before the descent when none of its queries touches the other tree's root
    box; with a partially built tree (built_level > 1) there is no root box and the exit must be skipped.  Disjoint,
    corner-touching and ragged (n % 64 != 0) clouds, contact cache on and off, both argument orders, the enqueue
    path — lists identical to the oracle's, order included."""
    from implicitbvh_amd import api
    monkeypatch.setattr(api, "LVT_CACHE_SLOTS", slots)
    rng = np.random.default_rng(31 + slots)
    types = abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32, idx, abi.U32)
    n1, n2 = 1000 + 37, 700 + 5
    a = random_volumes(rng, n1, abi.BSPHERE, abi.F32, scale=4.0, size=0.2)
    far = a[:n2].copy()
    far[:, :3] += 100.0                                   # disjoint: every wave leaves at once
    corner = random_volumes(rng, n2, abi.BSPHERE, abi.F32, scale=4.0, size=0.2)
    corner[:, :3] += 3.0                                  # only the corner region [3,4]^3 of `a` is in reach
    touch = corner.copy()
    touch[:, :3] += 0.95                                  # root boxes overlap by a sliver: no contacts at all
    for b in (far, corner, touch):
        for bl in (1, 3):
            (o1, g1), (o2, g2) = build_both(a, types, built_level=bl), build_both(b, types, built_level=bl)
            sl1, sl2 = max(bl, o1.tree.levels // 2), max(bl, o2.tree.levels // 2)
            e12 = oracle_pairs(orc.traverse_pair_lvt(o1, o2, sl1, sl2)[0])
            e21 = oracle_pairs(orc.traverse_pair_lvt(o2, o1, sl2, sl1)[0])
            t12 = ibvh.traverse(g1, g2, start_level1=sl1, start_level2=sl2)
            assert (contacts_np(t12) == e12).all() and len(e12) == t12.num_contacts
            t21 = ibvh.traverse(g2, g1, start_level1=sl2, start_level2=sl1, cache=t12)   # enqueue path, swapped
            assert (contacts_np(t21) == e21).all() and len(e21) == t21.num_contacts
            assert sorted(map(tuple, e12.tolist())) == sorted((y, x) for x, y in e21.tolist())
    assert len(oracle_pairs(orc.traverse_pair_lvt(*[build_both(v, types)[0] for v in (a, far)])[0])) == 0
    assert len(oracle_pairs(orc.traverse_pair_lvt(*[build_both(v, types)[0] for v in (a, corner)])[0])) > 0


def test_lvt_rays_identical_order_incl_zero_direction_components():
    rng = np.random.default_rng(23)
    for combo in ((abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), (abi.BBOX, abi.F64, abi.BBOX, abi.F64),
                  (abi.BSPHERE, abi.F64, abi.BSPHERE, abi.F64)):
        types = abi.make_types(*combo)
        f = NP_F[combo[1]]
        for n in (1, 7, 500):
            vols = random_volumes(rng, n, combo[0], combo[1])
            o, g = build_both(vols, types)
            p = (8 * rng.random((257, 3)) - 1).astype(f)
            d = (rng.random((257, 3)) - 0.5).astype(f)
            d[::7, 0] = 0
            d[::11, 1] = 0
            d[::13] = 0
            for sl in sorted({1, o.tree.levels}):
                exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d, sl)[0])
                got = ibvh.traverse_rays(g, cuda(p).t(), cuda(d).t(), start_level=sl)
                assert (contacts_np(got) == exp).all()
                bfs = ibvh.traverse_rays(g, cuda(p).t(), cuda(d).t(), ibvh.BFSTraversal(), start_level=sl)
                eb, res = orc.traverse_rays_bfs(o, p, d, sl)
                assert sorted(map(tuple, contacts_np(bfs).tolist())) == sorted(map(tuple, oracle_pairs(eb).tolist()))
                assert bfs.num_checks == res.num_checks
    with pytest.raises(ValueError):
        ibvh.traverse_rays(g, torch.zeros((2, 5)), torch.zeros((2, 5)))
    assert ibvh.traverse_rays(g, torch.zeros((3, 0)), torch.zeros((3, 0))).num_contacts == 0


def test_narrow_menu_bfs_equals_lvt():
    """runtests.jl:1230-1270 / gputests.jl:251-288 with narrow = (a, b) -> a.morton < b.morton."""
    rng = np.random.default_rng(24)
    types = abi.make_types()
    for n in (2, 43, 190, 2000):
        vols = random_volumes(rng, n, abi.BSPHERE, abi.F32)
        o, g = build_both(vols, types)
        exp = oracle_pairs(orc.traverse_lvt(o, narrow=abi.NARROW_MORTON_LT)[0])
        lvt = ibvh.traverse(g, narrow=ibvh.NARROW_MORTON_LT)
        bfs = ibvh.traverse(g, ibvh.BFSTraversal(), narrow=ibvh.NARROW_MORTON_LT)
        assert (contacts_np(lvt) == exp).all()
        assert sorted(map(tuple, contacts_np(bfs).tolist())) == sorted(map(tuple, exp.tolist()))
    with pytest.raises(TypeError):
        ibvh.traverse(g, narrow="morton")
    with pytest.raises(ValueError):
        ibvh.traverse(g, narrow=ibvh.NARROW_RAY_ORIGIN_OUTSIDE)  # a ray predicate is not on the pair menu


def _rays_positions(g, P_, D_):
    """traverse_rays with IBVH_OUTPUT_POSITIONS through the mirror's internals (no callable: the raw position list)"""
    from implicitbvh_amd import api
    saved = api._narrow_code
    try:
        api._narrow_code = lambda narrow, rays=False: (abi.OUTPUT_POSITIONS, None)
        return ibvh.traverse_rays(g, P_, D_)
    finally:
        api._narrow_code = saved


def _positions(leaves):
    """user index -> 1-based position in the sorted leaf array (indices are a permutation of 1..n in these tests)"""
    idx = leaves["index"].astype(np.int64)
    pos = np.zeros(idx.max() + 1, np.int64)
    pos[idx] = np.arange(1, len(idx) + 1)
    return pos


@pytest.mark.parametrize("combo", [ALL_COMBOS[0], ALL_COMBOS[1], ALL_COMBOS[4]], ids=str)
def test_contact_positions_option_and_callable_narrow(combo):
    """IBVH_OUTPUT_POSITIONS (include/ibvh.h): the same contact list with leaf positions instead of user indices, query /
    bvh1 first — what a host needs to evaluate ANY pure `narrow` itself (the reference only ever uses it as
    `iscontact(...) && narrow(...)` at leaf level: lvt/traverse_single.jl:170, bfs/traverse_single_gpu.jl:187).  The
    Python mirror does exactly that for a callable: it must give what the device menu gives for the same predicate
    (runtests.jl:1230-1270, gputests.jl:251-288), LVT in the reference's order, BFS as a set."""
    rng = np.random.default_rng(41)
    types = abi.make_types(*combo)
    f = NP_F[combo[1]]
    for n, n2 in ((2, 3), (61, 200), (3000, 1700)):
        v1, v2 = random_volumes(rng, n, combo[0], combo[1]), random_volumes(rng, n2, combo[0], combo[1])
        (o1, g1), (o2, g2) = build_both(v1, types), build_both(v2, types)
        P1, P2 = _positions(o1.leaves), _positions(o2.leaves)
        # self: (query position, partner position), query = the smaller position
        exp = oracle_pairs(orc.traverse_lvt(o1)[0]).astype(np.int64).reshape(-1, 2)
        want = np.stack([np.minimum(P1[exp[:, 0]], P1[exp[:, 1]]), np.maximum(P1[exp[:, 0]], P1[exp[:, 1]])], 1)
        for alg in (ibvh.LVTTraversal(), ibvh.BFSTraversal()):
            got = contacts_np(ibvh.api._traverse_lvt_single(g1, 1, abi.OUTPUT_POSITIONS, None) if isinstance(alg, ibvh.LVTTraversal)
                              else ibvh.api._traverse_bfs_single(g1, max(o1.tree.levels // 2, 1), abi.OUTPUT_POSITIONS, None)).reshape(-1, 2)
            if isinstance(alg, ibvh.LVTTraversal):
                assert (got == want).all()
            else:
                assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, want.tolist()))
        # pair: (position in bvh1, position in bvh2), both argument orders (the larger BVH drives: flip)
        for (oa, ga, Pa), (ob, gb, Pb) in (((o1, g1, P1), (o2, g2, P2)), ((o2, g2, P2), (o1, g1, P1))):
            exp = oracle_pairs(orc.traverse_pair_lvt(oa, ob)[0]).astype(np.int64).reshape(-1, 2)
            want = np.stack([Pa[exp[:, 0]], Pb[exp[:, 1]]], 1)
            got = contacts_np(ibvh.api._traverse_lvt_pair(ga, gb, 1, 1, abi.OUTPUT_POSITIONS, None)).reshape(-1, 2)
            assert (got == want).all()
            gotb = contacts_np(ibvh.api._traverse_bfs_pair(ga, gb, max(oa.tree.levels // 2, 1), max(ob.tree.levels // 2, 1),
                                                         abi.OUTPUT_POSITIONS, None)).reshape(-1, 2)
            assert sorted(map(tuple, gotb.tolist())) == sorted(map(tuple, want.tolist()))
        # a callable narrow == the device menu entry with the same meaning, and == the oracle
        for code, fn in ((ibvh.NARROW_MORTON_LT, lambda a, b: a.morton < b.morton), (ibvh.NARROW_INDEX_LT, lambda a, b: a.index < b.index)):
            exp = oracle_pairs(orc.traverse_lvt(o1, narrow=code)[0]).reshape(-1, 2)
            assert (contacts_np(ibvh.traverse(g1, narrow=fn)).reshape(-1, 2) == exp).all()
            assert (contacts_np(ibvh.traverse(g1, narrow=code)).reshape(-1, 2) == exp).all()
            bfs = contacts_np(ibvh.traverse(g1, ibvh.BFSTraversal(), narrow=fn)).reshape(-1, 2)
            assert sorted(map(tuple, bfs.tolist())) == sorted(map(tuple, exp.tolist()))
            expp = oracle_pairs(orc.traverse_pair_lvt(o1, o2, narrow=code)[0]).reshape(-1, 2)
            assert (contacts_np(ibvh.traverse(g1, g2, narrow=fn)).reshape(-1, 2) == expp).all()
            bfsp = contacts_np(ibvh.traverse(g1, g2, ibvh.BFSTraversal(), narrow=fn)).reshape(-1, 2)
            assert sorted(map(tuple, bfsp.tolist())) == sorted(map(tuple, expp.tolist()))
        # a predicate that is on no menu: volumes are visible to the callable
        if combo[0] == abi.BSPHERE:
            fn = lambda a, b: a.volume[:, 3] > b.volume[:, 3]  # noqa: E731  (query radius larger than the partner's)
            exp = oracle_pairs(orc.traverse_lvt(o1)[0]).astype(np.int64).reshape(-1, 2)
            rad = o1.leaves["volume"]["r"]
            qpos, ppos = np.minimum(P1[exp[:, 0]], P1[exp[:, 1]]), np.maximum(P1[exp[:, 0]], P1[exp[:, 1]])
            keep = rad[qpos - 1] > rad[ppos - 1]
            assert (contacts_np(ibvh.traverse(g1, narrow=fn)).reshape(-1, 2) == exp[keep]).all()
    with pytest.raises(ValueError):
        ibvh.traverse(g1, narrow=lambda a, b: a.index[:1] > 0)  # one bool per candidate is required


def test_ray_narrow_menu_and_positions():
    """traverse_rays(...; narrow) (raytrace/raytrace.jl:76, raytrace/leaf_vs_tree/leaf_vs_tree.jl:194: `isintersection(...) &&
    narrow(leaf, p, d)`): the device menu entry IBVH_NARROW_RAY_ORIGIN_OUTSIDE, a callable doing the same on the host side of
    the boundary, and the positions option — all against the oracle's hit list filtered by the predicate (a pure narrow is
    a post-filter)."""
    rng = np.random.default_rng(42)
    for combo in ((abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), (abi.BBOX, abi.F64, abi.BBOX, abi.F64)):
        types = abi.make_types(*combo)
        f = NP_F[combo[1]]
        for n in (1, 9, 1500):
            vols = random_volumes(rng, n, combo[0], combo[1], scale=4.0)
            o, g = build_both(vols, types)
            nr = 700
            p = (5 * rng.random((nr, 3)) - 0.5).astype(f)
            p[::3] = vols[rng.integers(0, n, len(p[::3])), :3].astype(f) if combo[0] == abi.BSPHERE else \
                (0.5 * (vols[rng.integers(0, n, len(p[::3])), :3] + vols[rng.integers(0, n, len(p[::3])), 3:])).astype(f)  # origins INSIDE leaves
            d = (rng.random((nr, 3)) - 0.5).astype(f)
            exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d)[0]).astype(np.int64).reshape(-1, 2)
            P = _positions(o.leaves)
            lv = o.leaves["volume"][P[exp[:, 0]] - 1]
            pp = p[exp[:, 1] - 1]
            if combo[0] == abi.BSPHERE:
                dd = pp - lv["x"]
                outside = dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1] + dd[:, 2] * dd[:, 2] > lv["r"] * lv["r"]
            else:
                outside = ((pp < lv["lo"]) | (pp > lv["up"])).any(axis=1)
            assert 0 < outside.sum() < len(exp) or n == 1
            P_, D_ = cuda(p).t(), cuda(d).t()
            got = contacts_np(ibvh.traverse_rays(g, P_, D_, narrow=ibvh.NARROW_RAY_ORIGIN_OUTSIDE)).reshape(-1, 2)
            assert (got == exp[outside]).all()
            bfs = contacts_np(ibvh.traverse_rays(g, P_, D_, ibvh.BFSTraversal(), narrow=ibvh.NARROW_RAY_ORIGIN_OUTSIDE)).reshape(-1, 2)
            assert sorted(map(tuple, bfs.tolist())) == sorted(map(tuple, exp[outside].tolist()))
            # cached (enqueue) path keeps the narrow
            t1 = ibvh.traverse_rays(g, P_, D_, narrow=ibvh.NARROW_RAY_ORIGIN_OUTSIDE)
            t2 = ibvh.traverse_rays(g, P_, D_, narrow=ibvh.NARROW_RAY_ORIGIN_OUTSIDE, cache=t1)
            assert (contacts_np(t2).reshape(-1, 2) == exp[outside]).all()
            # callable: hits whose ray index is even and whose leaf index is odd
            fn = lambda bv, pts, dirs: (bv.index % 2 == 1) & (pts[:, 0] == pts[:, 0])  # noqa: E731
            keep = exp[:, 0] % 2 == 1
            assert (contacts_np(ibvh.traverse_rays(g, P_, D_, narrow=fn)).reshape(-1, 2) == exp[keep]).all()
            bfs = contacts_np(ibvh.traverse_rays(g, P_, D_, ibvh.BFSTraversal(), narrow=fn)).reshape(-1, 2)
            assert sorted(map(tuple, bfs.tolist())) == sorted(map(tuple, exp[keep].tolist()))
    with pytest.raises(ValueError):
        ibvh.traverse_rays(g, P_, D_, narrow=ibvh.NARROW_MORTON_LT)  # a pair predicate is not on the ray menu


def test_work_counters_rays_equal_the_reference_walk():
    """ibvh_lvt_work_counters: the ray walker tests both children of every node it enters, exactly the nodes and leaves the
    reference's stack walk tests (raytrace/leaf_vs_tree/leaf_vs_tree.jl:187-225) -> identical counts; the BBox-node leaf
    walkers enumerate candidates differently (conservative union boxes), so there only sanity bounds hold."""
    n = 60_000
    r0 = 0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3)
    host = orc.generate_spheres_f32(n, 42, r0=r0)
    o, g = build_both(host, abi.make_types())
    rng = np.random.default_rng(3)
    p, d = rng.random((5000, 3)).astype(np.float32), (rng.random((5000, 3)) - 0.3).astype(np.float32)
    w = ibvh.lvt_work_counters(g, points=cuda(p).t(), directions=cuda(d).t())
    nt, lt, nc = orc.lvt_test_counts(o, points=p, directions=d)
    assert (w["node_tests"], w["leaf_tests"], w["contacts_counted"]) == (nt, lt, nc)
    assert w["node_fetches"] == nt and w["leaf_fetches"] == lt
    ws = ibvh.lvt_work_counters(g)
    nts, lts, ncs = orc.lvt_test_counts(o)
    assert ws["contacts_counted"] == ncs == ibvh.traverse(g).num_contacts
    assert ws["leaf_tests"] >= ncs and 0.5 * lts <= ws["leaf_tests"] <= 4 * lts
    assert ws["node_fetches"] < nts  # a wave shares its node fetches: far fewer records than one walk per leaf
    o2, g2 = build_both(orc.generate_spheres_f32(n, 45, origin=(0.9, 0, 0), r0=r0), abi.make_types())
    wp = ibvh.lvt_work_counters(g, g2)
    ntp, ltp, ncp = orc.lvt_test_counts(o, o2)
    assert wp["contacts_counted"] == ncp and wp["leaf_tests"] >= ncp


def test_more_contacts_than_int32_is_an_overflow_error():
    """IBVH_ERR_OVERFLOW (include/ibvh.h; SURVEY.md §8a trap 9: the reference has no guard, its Int32 prefix sum wraps):
    70,000 identical spheres touch pairwise — 2.45e9 > 2^31 - 1 contacts; the COUNT call must say so instead of returning
    a wrapped total, and the Python mirror raises OverflowError."""
    n = 70_000
    vols = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    vols[:, :3] = torch.rand((n, 3), device="cuda") * 1e-3
    vols[:, 3] = 1.0
    g = ibvh.BVH(vols)
    with pytest.raises(OverflowError):
        ibvh.traverse(g)
    # the same cloud with Int64 indices counts them all
    g64 = ibvh.BVH(vols, options=ibvh.BVHOptions(index=np.int64))
    s = g64.struct()
    need = C.c_size_t()
    lib.call("ibvh_lvt_scratch_bytes", C.byref(g64.types), n, 0, C.byref(need))
    scratch = torch.zeros(need.value, dtype=torch.uint8, device="cuda")
    counts = torch.zeros(n, dtype=torch.int64, device="cuda")
    total = C.c_int64()
    lib.call("ibvh_traverse_lvt_count", C.byref(s), 1, 0, counts.data_ptr(), C.byref(total), scratch.data_ptr(), scratch.numel(),
             torch.cuda.current_stream().cuda_stream)
    assert total.value == n * (n - 1) // 2


# ---------------------------------------------------------------------------------------------
# BFS: sorted-set equality + identical num_checks
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("combo", ALL_COMBOS[:5], ids=str)
def test_bfs_self_every_start_level(combo):
    rng = np.random.default_rng(31)
    types = abi.make_types(*combo)
    for n in (1, 2, 5, 12, 100, 1001):
        vols = random_volumes(rng, n, combo[0], combo[1])
        o, g = build_both(vols, types)
        for sl in range(1, o.tree.levels + 1):
            eb, res = orc.traverse_bfs(o, sl)
            got = ibvh.traverse(g, ibvh.BFSTraversal(), start_level=sl)
            assert sorted(map(tuple, contacts_np(got).tolist())) == sorted(map(tuple, oracle_pairs(eb).tolist())), (n, sl)
            assert got.num_checks == res.num_checks and got.num_contacts == res.num_contacts


def test_bfs_pair_six_phases():
    rng = np.random.default_rng(32)
    types = abi.make_types()
    for n1, n2 in ((1, 1), (1, 50), (50, 1), (22, 190), (190, 22), (64, 64), (3, 500), (500, 3)):
        a, b = random_volumes(rng, n1, abi.BSPHERE, abi.F32), random_volumes(rng, n2, abi.BSPHERE, abi.F32)
        o1, g1 = build_both(a, types)
        o2, g2 = build_both(b, types)
        for sl1 in range(1, o1.tree.levels + 1):
            for sl2 in range(1, o2.tree.levels + 1):
                eb, res = orc.traverse_pair_bfs(o1, o2, sl1, sl2)
                got = ibvh.traverse(g1, g2, ibvh.BFSTraversal(), start_level1=sl1, start_level2=sl2)
                assert sorted(map(tuple, contacts_np(got).tolist())) == sorted(map(tuple, oracle_pairs(eb).tolist()))
                assert got.num_checks == res.num_checks


def test_bfs_large_with_capacity_growth_and_cache():
    n = 100000
    sph = orc.generate_spheres_f32(n, 8, r0=0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3))
    o, g = build_both(sph, abi.make_types())
    eb, res = orc.traverse_bfs(o)
    got = ibvh.traverse(g, ibvh.BFSTraversal())
    assert got.num_checks == res.num_checks
    assert sorted(map(tuple, contacts_np(got).tolist())) == sorted(map(tuple, oracle_pairs(eb).tolist()))
    again = ibvh.traverse(g, ibvh.BFSTraversal(), cache=got)
    assert again.num_contacts == got.num_contacts
    lvt = ibvh.traverse(g)
    assert sorted(map(tuple, contacts_np(lvt).tolist())) == sorted(map(tuple, contacts_np(got).tolist()))


def test_bfs_resumes_at_the_level_that_overflowed(monkeypatch):
    """Queues that start at exactly the initial pair count and grow only to what the overflowed level needs: nearly every
    level overflows once, and every time the traversal must RESUME there (ibvh_bfs_result.resume_step / resume_num) — same
    contacts and the same num_checks as the oracle for one BVH, two BVHs and rays."""
    from implicitbvh_amd import api
    monkeypatch.setattr(api, "BFS_INITIAL_FACTOR", 1)
    monkeypatch.setattr(api, "BFS_GROWTH", 1)
    calls = {"n": 0, "resumed": 0}
    real = lib.load().ibvh_traverse_bfs

    rng = np.random.default_rng(41)
    types = abi.make_types()
    n = 30000
    sph = orc.generate_spheres_f32(n, 9, r0=0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3))
    o, g = build_both(sph, types)
    eb, res = orc.traverse_bfs(o)
    got = ibvh.traverse(g, ibvh.BFSTraversal())
    assert got.num_checks == res.num_checks and got.num_contacts == res.num_contacts
    assert sorted(map(tuple, contacts_np(got).tolist())) == sorted(map(tuple, oracle_pairs(eb).tolist()))
    # the raw ABI: count the resumes of one traversal
    s = g.struct()
    cap = C.c_int64()
    lib.call("ibvh_bfs_initial_capacity", C.byref(s), 8, C.byref(cap))
    q = [torch.empty((cap.value, 2), dtype=torch.int32, device="cuda") for _ in range(2)]
    need = C.c_size_t()
    lib.call("ibvh_bfs_counters_bytes", g.tree.levels, C.byref(need))
    counters = torch.zeros(need.value, dtype=torch.uint8, device="cuda")
    r = abi.BfsResult()
    resumes = 0
    while True:
        st = real(C.byref(s), 8, 0, q[0].data_ptr(), q[1].data_ptr(), min(q[0].shape[0], q[1].shape[0]), counters.data_ptr(),
                  C.byref(r), torch.cuda.current_stream().cuda_stream)
        if st != abi.ERR_CAPACITY:
            break
        assert r.resume_step >= resumes - 1 and r.resume_num > 0  # never back to the start
        resumes += 1
        grown = []
        for k in (1, 2):
            t_ = torch.empty((int(r.required_capacity), 2), dtype=torch.int32, device="cuda")
            if k == r.contacts_in:
                t_[: r.resume_num].copy_(q[k - 1][: r.resume_num])
            grown.append(t_)
        q = grown
    assert st == 0 and resumes >= 3
    e8, r8 = orc.traverse_bfs(o, 8)
    assert r.num_checks == r8.num_checks and r.num_contacts == r8.num_contacts
    # two BVHs and rays through the Python driver (same resume loop)
    b2 = random_volumes(rng, 5000, abi.BSPHERE, abi.F32, scale=10.0)
    o2, g2 = build_both(b2, types)
    b3 = random_volumes(rng, 7000, abi.BSPHERE, abi.F32, scale=10.0)
    o3, g3 = build_both(b3, types)
    ep, rp = orc.traverse_pair_bfs(o2, o3, 3, 4)
    gp = ibvh.traverse(g2, g3, ibvh.BFSTraversal(), start_level1=3, start_level2=4)
    assert gp.num_checks == rp.num_checks
    assert sorted(map(tuple, contacts_np(gp).tolist())) == sorted(map(tuple, oracle_pairs(ep).tolist()))
    p = rng.random((3000, 3)).astype(np.float32) * 10
    d = (rng.random((3000, 3)) - 0.5).astype(np.float32)
    er, rr = orc.traverse_rays_bfs(o2, p, d, 2)
    gr = ibvh.traverse_rays(g2, torch.from_numpy(p).cuda().t(), torch.from_numpy(d).cuda().t(), ibvh.BFSTraversal(), start_level=2)
    assert gr.num_checks == rr.num_checks
    assert sorted(map(tuple, contacts_np(gr).tolist())) == sorted(map(tuple, oracle_pairs(er).tolist()))


# ---------------------------------------------------------------------------------------------
# multi-GPU build, emulated with virtual ranks on this one GPU (the library's driver + in-process collectives)
# ---------------------------------------------------------------------------------------------
def _virtual_ranks():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import virtual_ranks
    return virtual_ranks



@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_distributed_build_virtual_ranks_gpu(world):
    from implicitbvh_amd import dist as ibd
    n = 200003
    r0 = 0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3)
    host = orc.generate_spheres_f32(n, 46, r0=r0)
    single = orc.build(host, abi.make_types())
    bounds = [n * r // world for r in range(world + 1)]

    def fn(comm):
        vols = ibvh.generate_spheres(bounds[comm.rank + 1] - bounds[comm.rank], 46, first_index=bounds[comm.rank], r0=r0)
        builder = ibd.DistributedBuilder(comm)
        bvh = builder.build(vols)
        trav = ibvh.traverse(bvh)
        torch.cuda.synchronize()
        return bvh.leaves.to_numpy(), bvh.nodes.cpu().numpy(), contacts_np(trav), builder.last
    out = _virtual_ranks().run_virtual_ranks(world, fn)
    cat = np.concatenate([o[0] for o in out])
    assert cat.tobytes() == single.leaves.tobytes()  # global stable-sorted sequence, global 1-based indices
    for leaves, nodes, contacts, last in out:
        assert last["extrema"].tobytes() == single.extrema.tobytes()
        # the per-slice tree and its self-contacts equal the oracle run on the same slice
        v = leaves["volume"]
        sl = np.concatenate([v["x"], v["r"][:, None]], axis=1)
        o = orc.build(sl, abi.make_types(), indices=leaves["index"], compute_extrema=False,
                      mins=single.extrema[:3], maxs=single.extrema[3:])
        assert nodes.tobytes() == o.nodes.tobytes()
        assert (contacts == oracle_pairs(orc.traverse_lvt(o)[0])).all()
    sizes = [len(o[0]) for o in out]
    assert max(sizes) - min(sizes) <= max(8, n // (50 * world))


@pytest.mark.parametrize("world,tolerance", [(3, 0.0), (4, 0.005)])
def test_distributed_build_clustered_duplicates_and_ragged_shards_gpu(world, tolerance):
    """The distributed build on input that defeats the one-level splitter search — clustered centres, many exact
    duplicates (equal keys straddle rank boundaries), ragged shard sizes including an EMPTY rank — with exact splitters
    (tolerance 0: refinement to full key resolution, count exchange, the tensor-op partition path) and the default:
    the concatenated slices must still be the single-device sorted array, bit for bit."""
    from implicitbvh_amd import dist as ibd
    rng = np.random.default_rng(9)
    n = 60000
    base = (rng.random((300, 3)) * 4).astype(np.float32)
    c = base[rng.integers(0, 300, n)] + (0.01 * rng.standard_normal((n, 3))).astype(np.float32)
    c[rng.random(n) < 0.3] = base[7]                     # 30 % exact duplicates of one point
    host = np.concatenate([c, (0.02 + 0.05 * rng.random((n, 1))).astype(np.float32)], axis=1)
    single = orc.build(host, abi.make_types())
    cuts = sorted(rng.integers(0, n, world - 1).tolist())
    bounds = [0] + cuts + [n]
    bounds[1] = bounds[0]                                # rank 0 holds nothing
    dev = cuda(host)

    def fn(comm):
        vols = dev[bounds[comm.rank]:bounds[comm.rank + 1]].contiguous()
        builder = ibd.DistributedBuilder(comm, tolerance=tolerance)
        bvh = builder.build(vols)
        torch.cuda.synchronize()
        leaves = bvh.leaves.to_numpy()
        leaves["index"] = leaves["index"]                # (global numbering follows the concatenated shards)
        return leaves, builder.last
    out = _virtual_ranks().run_virtual_ranks(world, fn)
    cat = np.concatenate([o[0] for o in out])
    for field in ("morton", "index"):
        assert cat[field].tolist() == single.leaves[field].tolist(), field
    assert cat["volume"].tobytes() == single.leaves["volume"].tobytes()
    if tolerance == 0.0:
        sizes = [len(o[0]) for o in out]
        assert max(sizes) - min(sizes) <= 1              # exact splitters: perfectly balanced


@pytest.mark.parametrize("world", [2, 5])
def test_cross_shard_completion_gives_the_global_contact_set(world):
    """per-slice self contacts + cross-slice pair contacts == contacts of the whole cloud."""
    from implicitbvh_amd import dist as ibd
    n = 60011
    r0 = 0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3)
    host = orc.generate_spheres_f32(n, 47, r0=r0)
    want = {tuple(p) for p in oracle_pairs(orc.traverse_lvt(orc.build(host, abi.make_types()))[0]).tolist()}
    bounds = [n * r // world for r in range(world + 1)]

    def fn(comm):
        vols = ibvh.generate_spheres(bounds[comm.rank + 1] - bounds[comm.rank], 47, first_index=bounds[comm.rank], r0=r0)
        builder = ibd.DistributedBuilder(comm)
        bvh = builder.build(vols)
        own = contacts_np(ibvh.traverse(bvh))
        cross = builder.cross_contacts(bvh).cpu().numpy().astype(np.int64)
        torch.cuda.synchronize()
        return own, cross
    out = _virtual_ranks().run_virtual_ranks(world, fn)
    got = set()
    total = 0
    for own, cross in out:
        total += len(own) + len(cross)
        got |= {tuple(p) for p in own.tolist()}
        got |= {(min(a, b), max(a, b)) for a, b in cross.tolist()}
    assert total == len(got)  # nothing reported twice
    assert got == want
    assert sum(len(c) for _, c in out) > 0


# ---------------------------------------------------------------------------------------------
# input preparation
# ---------------------------------------------------------------------------------------------
def test_triangle_volumes_bit_exact():
    rng = np.random.default_rng(41)
    for flt, tdt in ((abi.F32, torch.float32), (abi.F64, torch.float64)):
        tris = (6 * rng.random((5000, 1, 3)) + rng.random((5000, 3, 3))).astype(NP_F[flt])
        tris[::50, 2] = tris[::50, 1] + (tris[::50, 1] - tris[::50, 0])  # collinear -> degenerate branch
        for kind, tok in ((abi.BSPHERE, ibvh.BSphere), (abi.BBOX, ibvh.BBox)):
            exp = orc.volumes_from_triangles(kind, flt, tris.reshape(-1, 9))
            got = ibvh.bounding_volumes_from_triangles(cuda(tris), tok(tdt))
            assert got.cpu().numpy().tobytes() == exp.tobytes()


def test_obj_ingest_to_contacts(tmp_path):
    """OBJ -> triangles -> bounding spheres -> BVH -> contacts, against the oracle on the same triangles."""
    obj = tmp_path / "quad_strip.obj"
    lines = ["# strip of quads"]
    nx = 40
    for i in range(nx + 1):
        lines += [f"v {i * 0.5} 0 {0.1 * (i % 3)}", f"v {i * 0.5} 1 {0.05 * (i % 5)}"]
    for i in range(nx):
        a, b, c, d = 2 * i + 1, 2 * i + 2, 2 * i + 4, 2 * i + 3
        lines.append(f"f {a}/1/1 {b}/2/2 {c}/3/3 {d}/4/4" if i % 2 else f"f {a} {b} {c} {d}")
    lines.append("f -1 -2 -3")
    obj.write_text("\n".join(lines))
    tris = ibvh.load_obj_triangles(str(obj))
    assert tris.shape == (2 * nx + 1, 3, 3)
    vols = ibvh.bounding_volumes_from_triangles(tris)
    host = tris.cpu().numpy().reshape(-1, 9)
    exp = orc.volumes_from_triangles(abi.BSPHERE, abi.F32, host)
    assert vols.cpu().numpy().tobytes() == exp.tobytes()
    g = ibvh.BVH(vols)
    o = orc.build(vols.cpu().numpy(), abi.make_types())
    assert (contacts_np(ibvh.traverse(g)) == oracle_pairs(orc.traverse_lvt(o)[0])).all()


# ---------------------------------------------------------------------------------------------
# full-size properties (BASELINE.json config 2: 1e6 leaves)
# ---------------------------------------------------------------------------------------------
def test_config2_one_million_properties():
    n = 1_000_000
    r0 = 0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3)
    vols = ibvh.generate_spheres(n, 42, r0=r0)
    g = ibvh.BVH(vols)
    leaves = g.leaves.to_numpy()
    m = leaves["morton"].astype(np.int64)
    assert (np.diff(m) >= 0).all()  # sortedness
    idx = leaves["index"].astype(np.int64)
    assert np.array_equal(np.sort(idx), np.arange(1, n + 1))  # a permutation of 1..n
    ties = m[1:] == m[:-1]
    assert (idx[1:][ties] > idx[:-1][ties]).all()  # stability: ties in input order
    host = vols.cpu().numpy()
    assert leaves["volume"].tobytes() == host[idx - 1].tobytes()  # records follow their index
    t = ibvh.traverse(g)
    c = contacts_np(t)
    assert (c[:, 0] < c[:, 1]).all() and len(np.unique(c, axis=0)) == len(c)
    # every reported pair really touches (exact reference predicate in float32)
    a, b = host[c[:, 0] - 1], host[c[:, 1] - 1]
    dx = a[:, :3] - b[:, :3]
    d2 = (dx[:, 0] * dx[:, 0] + dx[:, 1] * dx[:, 1]) + dx[:, 2] * dx[:, 2]
    rr = a[:, 3] + b[:, 3]
    assert (d2 <= rr * rr).all()
    # and the whole list equals the oracle's, order included (the oracle needs a few seconds at 1e6)
    o = orc.build(host, abi.make_types())
    exp = oracle_pairs(orc.traverse_lvt(o)[0])
    assert (c == exp).all()
    idem = ibvh.traverse(g, cache=t)
    assert (contacts_np(idem) == c).all()  # idempotence with cache reuse


def test_concurrent_builds_and_traversals_on_two_streams_from_two_threads():
    """Re-entrancy (include/ibvh.h: 'no global state'): two host threads, one HIP stream each, build and traverse
    different clouds at the same time, per-launch profiling switched on (its record list is shared); each thread's
    results must equal the oracle's."""
    import threading
    rng = np.random.default_rng(77)
    types = abi.make_types()
    clouds = [random_volumes(rng, n, abi.BSPHERE, abi.F32, scale=s) for n, s in ((30011, 25.0), (50021, 30.0))]
    oracles = [orc.build(c, types) for c in clouds]
    expect = [oracle_pairs(orc.traverse_lvt(o)[0]) for o in oracles]
    errors = []

    def work(k):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                dev = torch.from_numpy(clouds[k]).cuda()
                for _ in range(6):
                    g = ibvh.BVH(dev)
                    t = ibvh.traverse(g)
                    assert g.leaves.to_numpy().tobytes() == oracles[k].leaves.tobytes()
                    assert (contacts_np(t) == expect[k]).all()
            st.synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    lib.call("ibvh_profile_enable", 1)
    try:
        threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        cnt = C.c_int64()
        lib.call("ibvh_profile_count", C.byref(cnt))
        lib.call("ibvh_profile_enable", 0)
    assert not errors, errors
    assert cnt.value > 0


def test_pair_lvt_smaller_bvh_drives_option():
    """IBVH_PAIR_SMALLER_DRIVES (round 6; the cross-shard completion uses it): the BVH with FEWER leaves supplies the work items.
    Same pairs in (bvh1, bvh2) order as the reference's rule gives (lvt/traverse_pair.jl:15-36: the larger one drives), listed
    in the smaller BVH's leaf order; both argument orders."""
    from implicitbvh_amd import api
    rng = np.random.default_rng(5)
    big = np.concatenate([rng.random((30_000, 3)), 0.02 * rng.random((30_000, 1))], axis=1).astype(np.float32)
    small = np.concatenate([0.4 + 0.2 * rng.random((700, 3)), 0.03 * rng.random((700, 1))], axis=1).astype(np.float32)
    b_big, b_small = ibvh.BVH(cuda(big)), ibvh.BVH(cuda(small))
    for b1, b2 in ((b_big, b_small), (b_small, b_big)):
        sl1, sl2 = ibvh.default_start_level(b1), ibvh.default_start_level(b2)
        ref = ibvh.traverse(b1, b2).contacts.cpu().numpy()
        got = api._traverse_lvt_pair(b1, b2, sl1, sl2, abi.PAIR_SMALLER_DRIVES, None).contacts.cpu().numpy()
        assert len(ref) > 1000 and got.shape == ref.shape
        assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, ref.tolist()))
        # ordered by the smaller BVH's leaves: its side of the pairs follows its sorted leaf order
        col = 1 if b1 is b_big else 0
        pos = np.empty(len(b_small.leaves) + 1, dtype=np.int64)
        pos[b_small.leaves.index.cpu().numpy()] = np.arange(len(b_small.leaves))
        p = pos[got[:, col]]
        assert (p[1:] >= p[:-1]).all()
        # the cached (enqueue) path too
        t0 = api._traverse_lvt_pair(b1, b2, sl1, sl2, abi.PAIR_SMALLER_DRIVES, None)
        again = api._traverse_lvt_pair(b1, b2, sl1, sl2, abi.PAIR_SMALLER_DRIVES, t0).contacts.cpu().numpy()
        assert again.tobytes() == got.tobytes()
