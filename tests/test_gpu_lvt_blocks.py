"""The shared descent of the LVT counting pass (csrc/ibvh_lvt.hpp "BlockRows", lvt_block_frontier_kernel): one frontier per block
of consecutive sorted leaves, the waves of the block start at the cut level.  By default it engages from 2^17 work items;
here the knob lvt_blocks_min_items = 1 forces it onto trees the oracle walks in seconds, and the lists must be the oracle's IN
ORDER for every block size, both descents of the block kernel (one / two levels per trip), self and pair walks, ragged trees,
partially built trees, both index types, Float64 volumes, the narrow menu, positions, rows that overflow (the waves then
descend on their own), leaves with NaN / infinite components (the containment check), and a scratch without room for rows."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as orc
from test_gpu_parity import ALL_COMBOS, _positions, build_both, contacts_np, cuda, oracle_pairs, random_volumes

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi, lib  # noqa: E402

BOX_COMBOS = [c for c in ALL_COMBOS if c[2] == abi.BBOX]


@pytest.fixture
def forced_rows():
    """rows for every tree the launch rule allows at all; the knobs go back to their defaults afterwards"""
    keys = ("lvt_blocks", "lvt_blocks_min_items", "lvt_block_shift", "lvt_blocks_paired_below")
    saved = {}
    for k in keys:
        v = C.c_int32()
        lib.call("ibvh_get_tuning", k.encode(), C.byref(v))
        saved[k] = v.value
    lib.set_tuning("lvt_blocks_min_items", 1)
    yield lib.set_tuning
    for k, v in saved.items():
        lib.set_tuning(k, v)


def _kernels_of(fn):
    """kernel names the library launched inside fn()"""
    lib.call("ibvh_profile_enable", 1)
    fn()
    torch.cuda.synchronize()
    cnt = C.c_int64()
    lib.call("ibvh_profile_count", C.byref(cnt))
    names = []
    for i in range(cnt.value):
        name, ms = C.c_char_p(), C.c_float()
        lib.call("ibvh_profile_get", i, C.byref(name), C.byref(ms))
        names.append(name.value.decode())
    lib.call("ibvh_profile_enable", 0)
    return names


@pytest.mark.parametrize("shift", [9, 10, 11, 12])
@pytest.mark.parametrize("paired", [0, 1 << 30], ids=["one-level-trips", "two-level-trips"])
def test_self_and_pair_lists_in_order_for_every_block_size(forced_rows, shift, paired):
    forced_rows("lvt_block_shift", shift)
    forced_rows("lvt_blocks_paired_below", paired)
    rng = np.random.default_rng(100 + shift)
    types = abi.make_types()
    for n in (4097, 20_000, 65_537, 150_001):  # ragged last blocks; 17 .. 19 levels
        vols = random_volumes(rng, n, abi.BSPHERE, abi.F32, scale=0.9 * n ** (1 / 3))
        o, g = build_both(vols, types)
        names = _kernels_of(lambda: ibvh.traverse(g))
        assert any("lvt_block_frontier_kernel" in k for k in names) == (o.tree.levels - 7 - 7 >= 2 and o.tree.levels - shift >= 1), (n, names)
        exp = oracle_pairs(orc.traverse_lvt(o)[0])
        t = ibvh.traverse(g)
        assert (contacts_np(t) == exp).all(), n
        assert (contacts_np(ibvh.traverse(g, cache=t)) == exp).all(), n      # the enqueue path
        other = random_volumes(rng, n // 3 + 5, abi.BSPHERE, abi.F32, scale=0.9 * n ** (1 / 3))
        o2, g2 = build_both(other, types)
        assert (contacts_np(ibvh.traverse(g, g2)) == oracle_pairs(orc.traverse_pair_lvt(o, o2)[0])).all(), n
        assert (contacts_np(ibvh.traverse(g2, g)) == oracle_pairs(orc.traverse_pair_lvt(o2, o)[0])).all(), n  # flipped: g drives again


@pytest.mark.parametrize("combo", BOX_COMBOS, ids=str)
def test_every_box_node_combination_both_index_types_narrow_and_positions(forced_rows, combo):
    rng = np.random.default_rng(7)
    n = 40_000
    vols = random_volumes(rng, n, combo[0], combo[1], scale=22.0)
    for it, mt in ((abi.I32, abi.U32), (abi.I64, abi.U64)):
        types = abi.make_types(*combo, it, mt)
        o, g = build_both(vols, types)
        assert any("lvt_block_frontier_kernel" in k for k in _kernels_of(lambda: ibvh.traverse(g)))
        exp = oracle_pairs(orc.traverse_lvt(o)[0])
        assert (contacts_np(ibvh.traverse(g)) == exp).all()
        for nar in (ibvh.NARROW_MORTON_LT, ibvh.NARROW_INDEX_LT):
            assert (contacts_np(ibvh.traverse(g, narrow=nar)) == oracle_pairs(orc.traverse_lvt(o, narrow=nar)[0])).all()
        got = contacts_np(ibvh.api.traverse(g, narrow=lambda a, b: a.index > 0))  # a callable goes through IBVH_OUTPUT_POSITIONS
        assert (got == exp).all()


@pytest.mark.parametrize("built_level", [3, 8, 9])
def test_partially_built_trees(forced_rows, built_level):
    """built_level > 7 moves the start level of both descents (and, at 9, leaves fewer than two levels above the cut of an
    18-level tree: no rows); the block level must exist in the query-side tree."""
    rng = np.random.default_rng(11)
    n = 100_000
    types = abi.make_types()
    vols = random_volumes(rng, n, abi.BSPHERE, abi.F32, scale=40.0)
    o, g = build_both(vols, types, built_level=built_level)
    for sl in (built_level, built_level + 2, o.tree.levels - 1):
        exp = oracle_pairs(orc.traverse_lvt(o, sl)[0])
        assert (contacts_np(ibvh.traverse(g, start_level=sl)) == exp).all(), sl
    o2, g2 = build_both(random_volumes(rng, 30_000, abi.BSPHERE, abi.F32, scale=40.0), types, built_level=2)
    exp = oracle_pairs(orc.traverse_pair_lvt(o, o2, built_level, 2)[0])
    assert (contacts_np(ibvh.traverse(g, g2, start_level1=built_level, start_level2=2)) == exp).all()


def test_rows_that_overflow_fall_back_to_the_waves_own_descent(forced_rows):
    """Heavily overlapping leaves: a block's box touches far more than a row's 496 cut-level nodes, the row says -1 and every
    wave of the block descends on its own (and overflows ITS frontier into the exact walk where it must)."""
    rng = np.random.default_rng(5)
    n = 300_000
    vols = random_volumes(rng, n, abi.BSPHERE, abi.F32, scale=30.0, size=0.6)
    vols[::1000, 3] = 8.0  # a few huge spheres: their blocks touch everything
    o, g = build_both(vols, abi.make_types())
    assert o.tree.levels - 7 >= 7 + 2
    exp = oracle_pairs(orc.traverse_lvt(o)[0])
    t = ibvh.traverse(g)
    assert t.num_contacts == len(exp)
    assert (contacts_np(t) == exp).all()


def _keys(c):
    c = torch.as_tensor(c).long()
    return c[:, 0] * (1 << 32) + c[:, 1]


@pytest.mark.parametrize("what", ["nan", "inf", "both"])
def test_nan_and_infinite_radii(forced_rows, what):
    """Leaves whose boxes are NaN (r = NaN) or infinite (r = Inf).  merge.jl's `a < b ? a : b` drops a NaN on its left and keeps one
    on its right, so (i) a block's box need not contain its NaN-free leaves — every wave checks the containment of its valid
    queries and descends on its own otherwise — and (ii) a NaN node box prunes, in the reference's walk, leaves that walker 2
    (which tests the levels from 7 to the cut level and then the leaf parents, include/ibvh.h "NaN") still reaches: with NaN
    boxes in the tree its list is a SUPERSET of the reference's, with or without rows (round 1's behaviour, now stated and
    pinned).  Infinite boxes alone change nothing: rows on == rows off == the oracle, in order.  No pair is ever reported twice."""
    rng = np.random.default_rng(9)
    n = 70_000
    vols = random_volumes(rng, n, abi.BSPHERE, abi.F32, scale=36.0)
    bad = rng.choice(n, 400, replace=False)
    # (only radii: a NaN or infinite CENTRE has no Morton code to speak of, which is the build's business, not this test's)
    if what in ("nan", "both"):
        vols[bad[:200], 3] = np.nan       # x - r = x + r = NaN: a box that touches nothing
    if what in ("inf", "both"):
        vols[bad[200:], 3] = np.inf       # (-inf, +inf): touches everything that is a number
    types = abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32)
    with np.errstate(all="ignore"):
        o, g = build_both(vols, types)
        assert g.leaves.to_numpy()["index"].tolist() == o.leaves["index"].tolist()
        # (the same numbers in the same places; a NaN's sign and payload are the platform's)
        assert np.array_equal(g.nodes.cpu().numpy().view(np.float32).ravel(), o.nodes.view(np.float32).ravel(), equal_nan=True)
        exp = oracle_pairs(orc.traverse_lvt(o)[0])
    with_rows = ibvh.traverse(g).contacts.clone()
    forced_rows("lvt_blocks", 0)
    without = ibvh.traverse(g).contacts.clone()
    kw, kn, ko = _keys(with_rows), _keys(without), _keys(exp).cuda()
    assert torch.unique(kw).shape[0] == kw.shape[0] and torch.unique(kn).shape[0] == kn.shape[0]   # nothing twice
    if what == "inf":
        assert with_rows.shape[0] == len(exp) and (with_rows.cpu().numpy().astype(np.int64) == exp).all()
        assert torch.equal(with_rows, without)
    else:
        assert bool(torch.isin(ko, kw).all()) and bool(torch.isin(ko, kn).all())        # supersets of the reference's list
        assert bool(torch.isin(kn, kw).all())   # (a wave that takes its row never falls into the exact walk: rows find what no-rows finds, and more)
        if what == "nan":
            assert torch.equal(with_rows, without)


def test_a_scratch_without_room_for_rows_is_served_without_them(forced_rows):
    """The rows live at the end of the caller's scratch when it was sized with ibvh_lvt_scratch_bytes; the two-call protocol
    with a scratch of just the scan's size (no contact cache, no rows) gives the same list."""
    rng = np.random.default_rng(13)
    n = 50_000
    vols = random_volumes(rng, n, abi.BSPHERE, abi.F32, scale=30.0)
    types = abi.make_types()
    o, g = build_both(vols, types)
    exp = oracle_pairs(orc.traverse_lvt(o)[0])
    s = g.struct()
    full, none = C.c_size_t(), C.c_size_t()
    lib.call("ibvh_lvt_scratch_bytes", C.byref(g.types), n, 8, C.byref(full))
    lib.call("ibvh_lvt_scratch_bytes", C.byref(g.types), n, 0, C.byref(none))
    assert full.value > none.value > 0
    rows_bytes = -(-n // 512) * 2048  # (csrc/ibvh_lvt.hpp: blk_rows_bytes at BLK_SHIFT_MIN)
    qidx_bytes = -(-n * 4 // 256) * 256  # (the dense copy of the work items' indices the counting pass leaves for the writing pass)
    for nbytes, rows in ((full.value, True), (none.value, True), (none.value - qidx_bytes, True),
                         (none.value - qidx_bytes - rows_bytes - 512, False)):
        counts = torch.zeros(n, dtype=torch.int32, device="cuda")
        scratch = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        total = C.c_int64()
        names = _kernels_of(lambda: lib.call("ibvh_traverse_lvt_count", C.byref(s), 1, 0, counts.data_ptr(), C.byref(total),
                                             scratch.data_ptr(), nbytes, None))
        assert any("lvt_block_frontier_kernel" in k for k in names) == rows, nbytes
        assert total.value == len(exp)
        out = torch.zeros((total.value, 2), dtype=torch.int32, device="cuda")
        lib.call("ibvh_traverse_lvt_write", C.byref(s), 1, 0, counts.data_ptr(), out.data_ptr(), scratch.data_ptr(), nbytes, None)
        torch.cuda.synchronize()
        assert (out.cpu().numpy().astype(np.int64) == exp).all()
