"""The binned ray path (csrc/ibvh_lvt.hip "(3c)": the walk of raytrace/leaf_vs_tree/leaf_vs_tree.jl:170-228 cut at a
level, the bottom finished subtree by subtree out of LDS) against the oracle's walk: the SAME list in the SAME order.
The knob "rays_binned" = 2 forces the path onto trees far smaller than the ones it is chosen for, so that the oracle can
walk them: every F32 leaf / node kind, subtrees of 2 .. 1,024 leaves, ragged last levels, start levels up to the cut and
past it, irregular rays (zero / infinite / NaN components), the cached (enqueue) path, the ray narrow, positions, Int64
indices, and an item list that overflows (the stand-by binary walker must then serve the call)."""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi, api, lib  # noqa: E402

from test_gpu_parity import _positions, _rays_positions, build_both, contacts_np, cuda, oracle_pairs, random_volumes  # noqa: E402


class knobs:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            lib.set_tuning(k, v)
        api._shape_memo.clear()

    def __exit__(self, *exc):
        for k in self.kw:
            lib.set_tuning(k, {"rays_binned": 1}.get(k, 0))
        api._shape_memo.clear()


def _rays(rng, nr, extent):
    p = (rng.random((nr, 3)) * (extent + 3) - 1).astype(np.float32)
    d = (rng.random((nr, 3)) - 0.5).astype(np.float32)
    d[::5, rng.integers(0, 3)] = 0
    d[7::41] = 0
    d[3::53, 1] = np.inf
    d[11::67, 2] = 1e-45
    p[13::71, 0] = np.nan
    p[17::73, 1] = np.inf
    d[19::79] *= 1e30
    return p, d


COMBOS = [(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), (abi.BBOX, abi.F32, abi.BBOX, abi.F32),
          (abi.BSPHERE, abi.F32, abi.BSPHERE, abi.F32),
          (abi.BSPHERE, abi.F64, abi.BBOX, abi.F64), (abi.BBOX, abi.F64, abi.BBOX, abi.F64), (abi.BSPHERE, abi.F64, abi.BSPHERE, abi.F64)]
NP_F = {abi.F32: np.float32, abi.F64: np.float64}


@pytest.mark.parametrize("combo", COMBOS, ids=str)
@pytest.mark.parametrize("depth", [1, 3, 6, 10])
def test_binned_rays_identical_order(combo, depth):
    rng = np.random.default_rng(100 + depth)
    types = abi.make_types(*combo)
    with knobs(rays_binned=2, rays_subtree_depth=depth):
        for n in (5, 129, 1000, 4097, 40_001):
            vols = random_volumes(rng, n, combo[0], combo[1], scale=20.0 if n > 2000 else 8.0)
            if combo[0] == abi.BSPHERE and n > 100:  # infinite and huge leaves: infinite node boxes through the fast slab test
                vols[5::97, 3] = np.inf
                vols[11::89, 3] = 1e38
            o, g = build_both(vols, types)
            nr = 2500
            p, d = _rays(rng, nr, 20 if n > 2000 else 8)
            p, d = p.astype(NP_F[combo[1]]), d.astype(NP_F[combo[1]])  # (rays in the leaf float type, raytrace/lvt:116-125)
            P_, D_ = cuda(p).t(), cuda(d).t()
            for sl in sorted({1, 2, max(1, o.tree.levels - depth), o.tree.levels}):
                if sl > o.tree.levels:
                    continue
                with np.errstate(all="ignore"):
                    exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d, sl)[0]).reshape(-1, 2)
                t1 = ibvh.traverse_rays(g, P_, D_, start_level=sl)
                assert t1.num_contacts == len(exp), (n, sl)
                assert (contacts_np(t1).reshape(-1, 2) == exp).all(), (n, sl)
                t2 = ibvh.traverse_rays(g, P_, D_, start_level=sl, cache=t1)  # enqueue path into the cached buffer
                assert (contacts_np(t2).reshape(-1, 2) == exp).all(), (n, sl)
            with np.errstate(all="ignore"):
                exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d)[0]).reshape(-1, 2)
            # the ray narrow on the menu, and positions
            with np.errstate(all="ignore"):
                en = oracle_pairs(orc.traverse_rays_lvt(o, p, d)[0]).reshape(-1, 2)
            tn = ibvh.traverse_rays(g, P_, D_, narrow=ibvh.NARROW_RAY_ORIGIN_OUTSIDE)
            got = contacts_np(tn).reshape(-1, 2)
            assert set(map(tuple, got.tolist())) <= set(map(tuple, en.tolist()))
            lib.set_tuning("rays_binned", 0)
            api._shape_memo.clear()
            ref = contacts_np(ibvh.traverse_rays(g, P_, D_, narrow=ibvh.NARROW_RAY_ORIGIN_OUTSIDE)).reshape(-1, 2)
            lib.set_tuning("rays_binned", 2)
            api._shape_memo.clear()
            assert got.shape == ref.shape and (got == ref).all()  # (the binary walker's narrowed list is pinned to the oracle elsewhere)
            pos = _positions(o.leaves)
            raw = contacts_np(_rays_positions(g, P_, D_)).reshape(-1, 2)
            assert (raw[:, 0] == pos[exp[:, 0]]).all() and (raw[:, 1] == exp[:, 1]).all()


def test_binned_rays_overflowing_item_list_falls_back_to_the_walker():
    rng = np.random.default_rng(7)
    types = abi.make_types()
    vols = random_volumes(rng, 30_000, abi.BSPHERE, abi.F32, scale=10.0)
    o, g = build_both(vols, types)
    p, d = _rays(rng, 4000, 10)
    with np.errstate(all="ignore"):
        exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d)[0]).reshape(-1, 2)
    P_, D_ = cuda(p).t(), cuda(d).t()
    with knobs(rays_binned=2, rays_subtree_depth=4, rays_items_per_ray=1):  # far fewer slots than items: overflow
        t1 = ibvh.traverse_rays(g, P_, D_)
        assert (contacts_np(t1).reshape(-1, 2) == exp).all()
        t2 = ibvh.traverse_rays(g, P_, D_, cache=t1)
        assert (contacts_np(t2).reshape(-1, 2) == exp).all()
    with knobs(rays_binned=2, rays_subtree_depth=4, rays_items_per_ray=64):  # and with room: the binned path itself
        t3 = ibvh.traverse_rays(g, P_, D_)
        assert (contacts_np(t3).reshape(-1, 2) == exp).all()
    with knobs(rays_binned=2, rays_subtree_depth=11):  # 2,048-leaf subtrees: more than 64 KB of LDS a workgroup
        t4 = ibvh.traverse_rays(g, P_, D_)
        assert (contacts_np(t4).reshape(-1, 2) == exp).all()
        bt = abi.make_types(abi.BBOX, abi.F32, abi.BBOX, abi.F32, index_type=abi.I64)
        bv = random_volumes(rng, 30_000, abi.BBOX, abi.F32, scale=10.0)
        ob, gb = build_both(bv, bt)
        with np.errstate(all="ignore"):
            eb = oracle_pairs(orc.traverse_rays_lvt(ob, p, d)[0]).reshape(-1, 2)
        assert (contacts_np(ibvh.traverse_rays(gb, P_, D_)).reshape(-1, 2) == eb).all()


def test_binned_rays_int64_indices_and_the_default_rule():
    """Int64 contacts through the binned path, and the shipped rule (knob = 1): a tree of 17+ levels with enough rays takes
    it, a small batch does not — same list either way."""
    rng = np.random.default_rng(9)
    types = abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32, index_type=abi.I64)
    vols = random_volumes(rng, 70_000, abi.BSPHERE, abi.F32, scale=30.0)
    o, g = build_both(vols, types)
    assert o.tree.levels >= 17
    p, d = _rays(rng, 20_000, 30)
    with np.errstate(all="ignore"):
        exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d)[0]).reshape(-1, 2)
    P_, D_ = cuda(p).t(), cuda(d).t()
    for mode in (2, 1, 0):
        with knobs(rays_binned=mode):
            t = ibvh.traverse_rays(g, P_, D_)
            assert t.contacts.dtype == torch.int64
            assert (contacts_np(t).reshape(-1, 2) == exp).all(), mode
            few = ibvh.traverse_rays(g, P_[:, :100].contiguous(), D_[:, :100].contiguous())
            assert (contacts_np(few).reshape(-1, 2) == exp[exp[:, 1] <= 100]).all()


def test_binned_rays_partially_built_trees():
    """built_level > 1 (the node levels above it do not exist): the cut level never lies above it, the walk starts at or below it."""
    rng = np.random.default_rng(31)
    types = abi.make_types()
    vols = random_volumes(rng, 20_000, abi.BSPHERE, abi.F32, scale=12.0)
    p, d = _rays(rng, 3000, 12)
    P_, D_ = cuda(p).t(), cuda(d).t()
    for built_level, depth in ((3, 6), (9, 10), (12, 2), (15, 10)):
        o, g = build_both(vols, types, built_level=built_level)
        with knobs(rays_binned=2, rays_subtree_depth=depth):
            for sl in sorted({built_level, min(o.tree.levels, built_level + 2), o.tree.levels}):
                with np.errstate(all="ignore"):
                    exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d, sl)[0]).reshape(-1, 2)
                t = ibvh.traverse_rays(g, P_, D_, start_level=sl)
                assert (contacts_np(t).reshape(-1, 2) == exp).all(), (built_level, depth, sl)


@pytest.mark.parametrize("seed", range(8))
def test_binned_rays_seeds_under_the_shipped_rule(seed):
    """Random sizes on both sides of the shipped rule (17 levels = 65,537 leaves; 8,192 rays; two rays per leaf), random extents and ray
    mixes, default knobs: whichever path the rule picks, the oracle's list in the oracle's order."""
    rng = np.random.default_rng(1000 + seed)
    kind = (abi.BSPHERE, abi.BBOX)[seed % 2]
    types = abi.make_types(kind, abi.F32, abi.BBOX, abi.F32)
    n = int(rng.integers(40_000, 400_000))
    nr = int(rng.integers(2_000, 30_000))
    scale = float(rng.uniform(5.0, 60.0))
    vols = random_volumes(rng, n, kind, abi.F32, scale=scale, size=float(rng.uniform(0.2, 1.5)))
    o, g = build_both(vols, types)
    p, d = _rays(rng, nr, scale)
    if seed % 3 == 0:  # a batch of ordinary rays only: every wave takes the packed slab test
        p = (rng.random((nr, 3)) * scale).astype(np.float32)
        d = (rng.random((nr, 3)) - 0.5).astype(np.float32)
        d[d == 0] = 0.25
    with np.errstate(all="ignore"):
        exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d)[0]).reshape(-1, 2)
    P_, D_ = cuda(p).t(), cuda(d).t()
    t1 = ibvh.traverse_rays(g, P_, D_)
    assert t1.num_contacts == len(exp)
    assert (contacts_np(t1).reshape(-1, 2) == exp).all()
    t2 = ibvh.traverse_rays(g, P_, D_, cache=t1)
    assert (contacts_np(t2).reshape(-1, 2) == exp).all()


def _kernels_of(fn):
    """names of the library kernels one call of fn launches (the library's own launch profiler)"""
    import ctypes as C
    lib.call("ibvh_profile_enable", 1)
    try:
        fn()
        torch.cuda.synchronize()
        cnt = C.c_int64()
        lib.call("ibvh_profile_count", C.byref(cnt))
        names = set()
        for i in range(cnt.value):
            name, ms = C.c_char_p(), C.c_float()
            lib.call("ibvh_profile_get", i, C.byref(name), C.byref(ms))
            names.add(name.value.decode().strip("() ").split("<")[0].split("::")[-1].strip())
        return names
    finally:
        lib.call("ibvh_profile_enable", 0)


def test_the_shipped_rule_really_takes_the_path_it_names():
    """A regression in the scratch sizing or in the rule would be SILENT (the per-lane walker gives the same list): look at
    the kernels a call launches.  70,000 leaves (18 levels): 5,000 rays -> binned; 1e6 rays (more than two per leaf of a
    tree below 2^20 leaves) -> the per-lane walker; Float64 throughout -> binned; knob 0 -> the per-lane walker."""
    rng = np.random.default_rng(17)
    vols = random_volumes(rng, 70_000, abi.BSPHERE, abi.F32, scale=30.0)
    g = ibvh.BVH(cuda(vols))
    g64 = ibvh.BVH(cuda(vols.astype(np.float64)), ibvh.BBox(torch.float64))
    def rays(nr, dt):
        p = (rng.random((nr, 3)) * 30).astype(dt)
        d = (rng.random((nr, 3)) - 0.5).astype(dt)
        return cuda(p).t(), cuda(d).t()
    P, D = rays(5000, np.float32)
    k = _kernels_of(lambda: ibvh.traverse_rays(g, P, D))
    assert {"rays_top_kernel", "rays_subtree_kernel", "rays_place_kernel"} <= k
    Pm, Dm = rays(300_000, np.float32)
    k = _kernels_of(lambda: ibvh.traverse_rays(g, Pm, Dm))
    assert "rays_top_kernel" not in k and "lvt_rays_kernel" in k
    P6, D6 = rays(5000, np.float64)
    k = _kernels_of(lambda: ibvh.traverse_rays(g64, P6, D6))
    assert {"rays_top_kernel", "rays_subtree_kernel"} <= k
    with knobs(rays_binned=0):
        k = _kernels_of(lambda: ibvh.traverse_rays(g, P, D))
        assert "rays_top_kernel" not in k and "lvt_rays_kernel" in k


def _subtree_passes_ms(fn):
    """(counting pass, writing pass) durations of rays_subtree_kernel in one call of fn, in launch order"""
    import ctypes as C
    lib.call("ibvh_profile_enable", 1)
    try:
        fn()
        torch.cuda.synchronize()
        cnt = C.c_int64()
        lib.call("ibvh_profile_count", C.byref(cnt))
        out = []
        for i in range(cnt.value):
            name, ms = C.c_char_p(), C.c_float()
            lib.call("ibvh_profile_get", i, C.byref(name), C.byref(ms))
            if "rays_subtree_kernel" in name.value.decode():
                out.append(ms.value)
        return out
    finally:
        lib.call("ibvh_profile_enable", 0)


@pytest.mark.parametrize("n_leaves,n_rays", [(249_882, 100_000), (70_000, 1_000), (70_000, 64)])
def test_the_counting_pass_keeps_its_hits_for_the_writing_pass(n_leaves, n_rays):
    """The hit list of the counting pass (records placed by rays_place_kernel) must serve batches of every size: with 256 lists
    whatever the batch (rounds 4 - 5), 1e5 rays on the reference's published mesh size overflowed one list and a 1,000-ray batch
    had 63 records a list — the writing pass then walked every subtree again, silently (same list, twice the time).  Seen in the
    library's launch profile: the writing pass of rays_subtree_kernel returns at once when the list holds everything."""
    from implicitbvh_amd.synthetic import random_rays, torus_mesh
    u = int(np.sqrt(n_leaves / 2)) + 2
    tris = torus_mesh(u, u)[:n_leaves].reshape(-1, 3, 3)
    c = tris.mean(1)
    r = np.linalg.norm(tris - c[:, None, :], axis=2).max(1, keepdims=True)
    vols = np.concatenate([c, r], axis=1).astype(np.float32)
    g = ibvh.BVH(cuda(vols))
    o = orc.build(vols, abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32))
    p, d = random_rays(n_rays, vols[:, :3].min(0), vols[:, :3].max(0), seed=5)
    P, D = cuda(p).t(), cuda(d).t()
    t = ibvh.traverse_rays(g, P, D)  # (warm: sizes the cache)
    exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d, 1)[0])
    assert (contacts_np(t).reshape(-1, 2) == exp).all()
    best = None
    for _ in range(3):
        ms = _subtree_passes_ms(lambda: ibvh.traverse_rays(g, P, D, cache=t))
        assert len(ms) == 2, ms
        best = ms if best is None or ms[1] / ms[0] < best[1] / best[0] else best
    assert best[1] < 0.3 * best[0] + 0.01, f"the writing pass walked again: count {best[0]:.3f} ms, write {best[1]:.3f} ms"
