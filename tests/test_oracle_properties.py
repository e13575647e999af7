"""Property tests of the CPU oracle that mirror the reference's randomised tests
(test/runtests.jl:839-1081, 1230-1270): BVH contacts == O(n^2) brute force for every start level,
traverse(bvh, bvh) symmetry, BFS == LVT under a narrow function.  Julia's Random.seed!(42) stream
cannot be regenerated, so the inputs are numpy's; the properties are the reference's."""
import numpy as np
import pytest

import oracle_lib as orc
from implicitbvh_amd import abi


def random_spheres(rng, n, flt=np.float64):
    # map(BSphere, [6 * rand(3) .+ rand(3, 3) ...]) — runtests.jl:849: spheres of random triangles
    tris = (6 * rng.random((n, 1, 3)) + rng.random((n, 3, 3))).astype(flt)
    s = orc.volumes_from_triangles(abi.BSPHERE, abi.F64 if flt == np.float64 else abi.F32, tris.reshape(n, 9))
    return np.concatenate([s["x"], s["r"][:, None]], axis=1)


def as_set(c):
    return sorted(orc.pairs_as_tuples(c))


@pytest.mark.parametrize("alg", ["lvt", "bfs"])
@pytest.mark.parametrize("node", [(abi.BBOX, abi.F32), (abi.BBOX, abi.F64), (abi.BSPHERE, abi.F64)])
def test_single_vs_brute_force(alg, node):
    rng = np.random.default_rng(42)
    trav = orc.traverse_lvt if alg == "lvt" else orc.traverse_bfs
    for n in range(1, 200, 11):
        tree = orc.tree_shape(n)
        for start_level in range(1, tree.levels + 1):
            sph = random_spheres(rng, n)
            types = abi.make_types(abi.BSPHERE, abi.F64, node[0], node[1])
            bvh = orc.build(sph, types)
            got = as_set(trav(bvh, start_level)[0])
            brute = sorted(map(tuple, orc.brute_force_self(abi.BSPHERE, abi.F64, sph).tolist()))
            assert got == brute, (n, start_level)


@pytest.mark.parametrize("alg", ["lvt", "bfs"])
def test_pair_equivalent_to_self(alg):
    """runtests.jl:936-1004: traverse(bvh, bvh) == self-contacts + diagonal + mirrored."""
    rng = np.random.default_rng(7)
    for n in range(1, 200, 33):
        tree = orc.tree_shape(n)
        sph = random_spheres(rng, n)
        bvh = orc.build(sph, abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F64))
        for sl1 in range(1, tree.levels + 1):
            for sl2 in range(1, tree.levels + 1):
                if alg == "lvt":
                    c1 = as_set(orc.traverse_lvt(bvh, sl1)[0])
                    c2 = orc.pairs_as_tuples(orc.traverse_pair_lvt(bvh, bvh, sl1, sl2)[0])
                else:
                    c1 = as_set(orc.traverse_bfs(bvh, sl1)[0])
                    c2 = orc.pairs_as_tuples(orc.traverse_pair_bfs(bvh, bvh, sl1, sl2)[0])
                s2 = set(c2)
                assert len(s2) == len(c2)
                assert all((i, i) in s2 for i in range(1, n + 1))
                off = {(i, j) for (i, j) in s2 if i != j}
                assert all((j, i) in off for (i, j) in off)
                assert sorted((i, j) for (i, j) in off if i < j) == c1


@pytest.mark.parametrize("alg", ["lvt", "bfs"])
def test_pair_vs_brute_force(alg):
    rng = np.random.default_rng(3)
    for n1 in range(1, 200, 41):
        for n2 in range(1, 200, 41):
            t1, t2 = orc.tree_shape(n1), orc.tree_shape(n2)
            for sl1 in range(1, t1.levels + 1):
                for sl2 in range(1, t2.levels + 1):
                    a, b = random_spheres(rng, n1), random_spheres(rng, n2)
                    types = abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F32)
                    b1, b2 = orc.build(a, types), orc.build(b, types)
                    trav = orc.traverse_pair_lvt if alg == "lvt" else orc.traverse_pair_bfs
                    got = as_set(trav(b1, b2, sl1, sl2)[0])
                    brute = sorted(map(tuple, orc.brute_force_pair(abi.BSPHERE, abi.F64, a, b).tolist()))
                    assert got == brute, (n1, n2, sl1, sl2)


def test_rays_vs_brute_force():
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 33, 150):
        sph = random_spheres(rng, n, np.float32)
        bvh = orc.build(sph, abi.make_types())
        p = (8 * rng.random((64, 3)) - 1).astype(np.float32)
        d = (rng.random((64, 3)) - 0.5).astype(np.float32)
        d[::7, 0] = 0.0  # zero direction components: inf/NaN slab paths (isintersection.jl:7-32)
        d[::11, 1] = 0.0
        brute = sorted(map(tuple, orc.brute_force_rays(abi.BSPHERE, abi.F32, sph, p, d).tolist()))
        for sl in range(1, bvh.tree.levels + 1):
            lvt = as_set(orc.traverse_rays_lvt(bvh, p, d, sl)[0])
            bfs = as_set(orc.traverse_rays_bfs(bvh, p, d, sl)[0])
            assert lvt == bfs
            # the BVH may only LOSE hits where the (non-conservative) slab test on nodes rejects a
            # ray that grazes; every BVH hit must be a brute-force hit
            assert set(lvt) <= set(brute)
        assert as_set(orc.traverse_rays_lvt(bvh, p, d, bvh.tree.levels)[0]) == brute


def test_narrow_bfs_equals_lvt():
    """runtests.jl:1230-1270."""
    rng = np.random.default_rng(11)
    for n in range(1, 200, 21):
        sph = random_spheres(rng, n)
        bvh = orc.build(sph, abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F32))
        a = as_set(orc.traverse_bfs(bvh, narrow=abi.NARROW_MORTON_LT)[0])
        b = as_set(orc.traverse_lvt(bvh, narrow=abi.NARROW_MORTON_LT)[0])
        assert a == b
    for n1 in range(1, 200, 61):
        for n2 in range(1, 200, 61):
            t = abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F32)
            b1, b2 = orc.build(random_spheres(rng, n1), t), orc.build(random_spheres(rng, n2), t)
            a = as_set(orc.traverse_pair_bfs(b1, b2, narrow=abi.NARROW_MORTON_LT)[0])
            b = as_set(orc.traverse_pair_lvt(b1, b2, narrow=abi.NARROW_MORTON_LT)[0])
            assert a == b


def test_extrema_strictly_bound_centres():
    """runtests.jl:510-559 incl. the 1- and 2-element degenerate inputs."""
    rng = np.random.default_rng(1)
    for flt, npf, scale in ((abi.F32, np.float32, 1000), (abi.F64, np.float64, 1000)):
        sph = (scale * rng.random((100, 4))).astype(npf)
        t = abi.make_types(abi.BSPHERE, flt, abi.BBOX, abi.F32)
        e = orc.extrema(t, orc.as_volumes(sph, abi.BSPHERE, flt), wrapped=False)
        assert (sph[:, :3] > e[:3]).all() and (sph[:, :3] < e[3:]).all()
    t = abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F32)
    for deg in ([[0, 0, 0, 1.0]], [[1000, 0, 0, 1.0], [1000, 0, 0, 1.0]]):
        sph = np.asarray(deg, np.float64)
        e = orc.extrema(t, orc.as_volumes(sph, abi.BSPHERE, abi.F64), wrapped=False)
        assert (sph[:, :3] > e[:3]).all() and (sph[:, :3] < e[3:]).all()


def test_built_level_and_fraction():
    tree = orc.tree_shape(100)
    assert orc.compute_build_level(tree, 0.0) == tree.levels
    assert orc.compute_build_level(tree, 1.0) == 1
    assert orc.compute_build_level(tree, 0.5) == round(tree.levels + (1 - tree.levels) * 0.5)
    rng = np.random.default_rng(2)
    sph = random_spheres(rng, 100)
    t = abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F64)
    full = orc.build(sph, t)
    part = orc.build(sph, t, built_level=3)
    lo = orc.memory_index(tree, 4)  # first node of level 3
    assert (part.nodes[lo - 1:] == full.nodes[lo - 1:]).all()
    c_full = as_set(orc.traverse_lvt(full, 3)[0])
    assert as_set(orc.traverse_lvt(part, 3)[0]) == c_full
    with pytest.raises(ValueError):
        orc.traverse_lvt(part, 2)  # start_level < built_level (lvt/traverse_single.jl:10)


def test_mt_baseline_matches_single_thread():
    sph = orc.generate_spheres_f32(20000, 42, r0=0.02)
    bvh = orc.build(sph, abi.make_types())
    ref, _ = orc.traverse_lvt(bvh)
    for threads in (1, 4):
        b2, contacts, tb, tt = orc.bench_build_traverse_f32(sph, threads)
        assert (b2.leaves == bvh.leaves).all() and (b2.nodes == bvh.nodes).all()
        assert (contacts == ref).all()
