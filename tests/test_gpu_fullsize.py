"""Full-size runs of BASELINE.json configs 3 and 4 on one MI355X, checked through size-independent
properties (the oracle cannot walk these sizes in test time): every reported pair re-checked with the exact
reference predicate, no duplicates, order, and — on samples — exact equality with the oracle's walk of the full-size tree
and completeness against brute force."""
import math

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi  # noqa: E402


from implicitbvh_amd.synthetic import torus_mesh  # noqa: E402  (shared with bench.py and tools/)


def contacts_np(trav):
    return trav.contacts.cpu().numpy().astype(np.int64)


def test_config3_mesh_rays_full_size():
    tris = torus_mesh()
    n = len(tris)
    assert n > 7_100_000
    dev = torch.from_numpy(tris).cuda()
    vols = ibvh.bounding_volumes_from_triangles(dev)
    host_vols = vols.cpu().numpy()
    # spot-check the triangle kernel against the oracle on a slice
    exp = orc.volumes_from_triangles(abi.BSPHERE, abi.F32, tris[:100000])
    assert host_vols[:100000].tobytes() == exp.tobytes()
    bvh = ibvh.BVH(vols)
    leaves = bvh.leaves.to_numpy()
    assert (np.diff(leaves["morton"].astype(np.int64)) >= 0).all()
    nr = 1_000_000
    rng = np.random.default_rng(43)  # benchmark/bvh_rays.jl:36-38: uniform points and directions
    lo, hi = host_vols[:, :3].min(0), host_vols[:, :3].max(0)
    p = (lo + (hi - lo) * rng.random((nr, 3))).astype(np.float32)
    d = rng.random((nr, 3)).astype(np.float32)
    trav = ibvh.traverse_rays(bvh, torch.from_numpy(p).cuda().t(), torch.from_numpy(d).cuda().t())
    c = trav.contacts.cpu().numpy().astype(np.int64)
    assert trav.num_contacts == len(c) > nr
    # contacts are (leaf.index, iray), grouped by ray in ray order (two-pass LVT)
    assert (np.diff(c[:, 1]) >= 0).all()
    assert c[:, 0].min() >= 1 and c[:, 0].max() <= n
    key = c[:, 1] * (n + 1) + c[:, 0]
    assert len(np.unique(key)) == len(c)
    assert orc.count_bad_ray_hits(abi.BSPHERE, abi.F32, host_vols, p, d, c) == 0
    # EXACT equality with the reference walk on a sample of the rays: the oracle builds the same 7.2 M-leaf tree on the
    # host (bit-identical leaves and nodes are asserted) and walks the sampled rays (raytrace/leaf_vs_tree/
    # leaf_vs_tree.jl:170-228); the GPU's list restricted to those rays must be that list, in that order.
    o = orc.build(host_vols, abi.make_types())
    assert leaves.tobytes() == o.leaves.tobytes()
    assert bvh.nodes.cpu().numpy().tobytes() == o.nodes.tobytes()
    sample = np.sort(rng.choice(nr, 3000, replace=False))
    exp, _ = orc.traverse_rays_lvt(o, p[sample], d[sample])
    want = np.stack([exp["a"].astype(np.int64), sample[exp["b"].astype(np.int64) - 1] + 1], axis=1)
    got = c[np.isin(c[:, 1] - 1, sample)]
    assert got.shape == want.shape and (got == want).all()
    # and against first principles on a few rays: brute force over all leaves can only find MORE (a leaf whose node boxes
    # the reference-identical slab test rejects is missed by the reference too), never fewer
    few = sample[:16]
    bf = orc.brute_force_rays(abi.BSPHERE, abi.F32, host_vols, p[few], d[few])
    bf_set = {(int(a_), int(few[b_ - 1]) + 1) for a_, b_ in bf}
    assert {(int(a_), int(b_)) for a_, b_ in c[np.isin(c[:, 1] - 1, few)]} <= bf_set
    bfs = ibvh.traverse_rays(bvh, torch.from_numpy(p[:20000]).cuda().t(), torch.from_numpy(d[:20000]).cuda().t(), ibvh.BFSTraversal())
    lvt = ibvh.traverse_rays(bvh, torch.from_numpy(p[:20000]).cuda().t(), torch.from_numpy(d[:20000]).cuda().t())
    assert sorted(map(tuple, bfs.contacts.cpu().numpy().tolist())) == sorted(map(tuple, lvt.contacts.cpu().numpy().tolist()))


def test_config4_pair_two_5e6_clouds_full_size():
    n = 5_000_000
    r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
    a = ibvh.generate_spheres(n, 44, r0=r0)
    b = ibvh.generate_spheres(n, 45, origin=(0.9, 0.0, 0.0), r0=r0)  # 10 % overlap in x
    ha, hb = a.cpu().numpy(), b.cpu().numpy()
    b1, b2 = ibvh.BVH(a), ibvh.BVH(b)
    trav = ibvh.traverse(b1, b2)
    c = trav.contacts.cpu().numpy().astype(np.int64)
    assert len(c) == trav.num_contacts > 100_000
    assert c.min() >= 1 and c.max() <= n
    assert len(np.unique(c[:, 0] * (n + 1) + c[:, 1])) == len(c)
    assert orc.count_bad_contacts(abi.BSPHERE, abi.F32, ha, hb, c) == 0
    # all contacts live in the overlap slab
    assert (ha[c[:, 0] - 1, 0] > 0.9 - 4 * r0).all() and (hb[c[:, 1] - 1, 0] < 1.0 + 4 * r0).all()
    # completeness: brute force for a sample of leaves of A that sit in the slab
    rng = np.random.default_rng(1)
    slab = np.nonzero(ha[:, 0] > 0.95)[0]
    sample = np.sort(rng.choice(slab, 64, replace=False))
    bf = orc.brute_force_pair(abi.BSPHERE, abi.F32, ha[sample], hb)
    want = {(int(sample[i - 1]) + 1, int(j)) for i, j in bf}
    got = {(int(i), int(j)) for i, j in c[np.isin(c[:, 0] - 1, sample)]}
    assert got == want  # BBox nodes of sphere leaves are conservative up to rounding: equality expected
    # symmetric call: (bvh2, bvh1) reports the mirrored pairs
    t2 = ibvh.traverse(b2, b1)
    c2 = t2.contacts.cpu().numpy().astype(np.int64)
    assert len(c2) == len(c)
    assert np.array_equal(np.unique(c2[:, 1] * (n + 1) + c2[:, 0]), np.unique(c[:, 0] * (n + 1) + c[:, 1]))
    # BFS agrees as a set
    bfs = ibvh.traverse(b1, b2, ibvh.BFSTraversal())
    cb = bfs.contacts.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.unique(cb[:, 0] * (n + 1) + cb[:, 1]), np.unique(c[:, 0] * (n + 1) + c[:, 1]))


def check_build_properties(vols, g):
    """Size-independent checks of a BVH of BSphere{Float32} leaves: codes ascending, .index a permutation, ties in
    input order, records follow their index, the codes are the oracle's for the build's extrema (on a sample), every
    level's nodes spot-checked as the exact merge of their children.  Returns (host volumes, leaves)."""
    n = len(g.leaves)
    leaves = g.leaves.to_numpy()
    m = leaves["morton"].astype(np.int64)
    assert (np.diff(m) >= 0).all()
    idx = leaves["index"].astype(np.int64)
    seen = np.zeros(n + 1, np.uint8)
    seen[idx] = 1
    assert seen[1:].all() and len(idx) == n                      # a permutation of 1..n
    ties = m[1:] == m[:-1]
    assert (idx[1:][ties] > idx[:-1][ties]).all()                # stable
    host = vols.cpu().numpy()
    assert leaves["volume"].tobytes() == host[idx - 1].tobytes()  # records follow their index
    rng = np.random.default_rng(3)
    sample = rng.integers(0, n, 100_000)
    keys = orc.morton_keys(abi.make_types(), np.ascontiguousarray(host[idx[sample] - 1]), False, g.extrema.cpu().numpy())
    assert np.array_equal(np.asarray(keys).astype(np.int64), m[sample])  # the codes themselves
    # nodes: level l node i == merge(children); checked exactly on random samples of every level
    tree = orc.tree_shape(n)
    nodes = g.nodes.cpu().numpy()                                # (real_nodes - real_leaves, 6) float32 boxes
    lv = leaves["volume"]
    leaf_lo = lv["x"] - lv["r"][:, None]
    leaf_up = lv["x"] + lv["r"][:, None]
    for level in range(tree.levels - 1, 0, -1):
        first = orc.memory_index(tree, 2 ** (level - 1)) - 1
        nreal = 2 ** (level - 1) - (tree.virtual_leaves >> (tree.levels - level))
        pick = np.unique(rng.integers(0, nreal, size=min(nreal, 2000)))
        if level == tree.levels - 1:
            l, r = 2 * pick, np.minimum(2 * pick + 1, n - 1)
            lo = np.minimum(leaf_lo[l], leaf_lo[r])
            up = np.maximum(leaf_up[l], leaf_up[r])
        else:
            cfirst = orc.memory_index(tree, 2 ** level) - 1
            creal = 2 ** level - (tree.virtual_leaves >> (tree.levels - level - 1))
            l, r = cfirst + 2 * pick, cfirst + np.minimum(2 * pick + 1, creal - 1)
            lo = np.minimum(nodes[l, :3], nodes[r, :3])
            up = np.maximum(nodes[l, 3:], nodes[r, 3:])
        assert np.array_equal(nodes[first + pick, :3], lo) and np.array_equal(nodes[first + pick, 3:], up), level
    return host, leaves


def test_north_star_ten_million_properties():
    """The north-star size (1e7 leaves, one GPU) through size-independent properties: sortedness, permutation,
    stability, records follow their index, every node is the exact merge of its children (spot-checked per level),
    every reported pair touches and is reported once, per-leaf counts agree with the list, BFS finds the same SET of
    contacts, and a second traversal into the cached buffers is identical."""
    n = 10_000_000
    r0 = 0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3)
    vols = ibvh.generate_spheres(n, 42, r0=r0)
    g = ibvh.BVH(vols)
    host, leaves = check_build_properties(vols, g)
    t = ibvh.traverse(g)
    c = contacts_np(t)
    assert len(c) > n and (c[:, 0] < c[:, 1]).all()
    key = c[:, 0] * (n + 1) + c[:, 1]
    assert len(np.unique(key)) == len(key)                       # nothing reported twice
    a, b = host[c[:, 0] - 1], host[c[:, 1] - 1]
    dx = a[:, :3] - b[:, :3]
    d2 = (dx[:, 0] * dx[:, 0] + dx[:, 1] * dx[:, 1]) + dx[:, 2] * dx[:, 2]
    rr = a[:, 3] + b[:, 3]
    assert (d2 <= rr * rr).all()                                 # the reference's float32 predicate, exactly
    counts = t.cache2.cpu().numpy().astype(np.int64)[:n]         # inclusive prefix of the per-leaf counts
    assert counts[-1] == len(c) and (np.diff(counts) >= 0).all()
    again = ibvh.traverse(g, cache=t)
    assert again.num_contacts == len(c) and (contacts_np(again) == c).all()
    bfs = ibvh.traverse(g, ibvh.BFSTraversal())
    cb = contacts_np(bfs)
    assert len(cb) == len(c)
    assert np.array_equal(np.sort(cb[:, 0] * (n + 1) + cb[:, 1]), np.sort(key))



@pytest.mark.parametrize("shape", ["clusters", "one_cluster", "few_centres"])
def test_skewed_ten_million_leaf_builds_properties(shape):
    """1e7 leaves that crowd a few cells of the sort's Morton grid — segments of millions of records, workgroups that
    do several partition tiles in a row, terminal segments, every extra partition level — checked through
    size-independent properties (check_build_properties); the rebuild with cache= gives the same bytes."""
    n = 10_000_000
    g = torch.Generator(device="cuda").manual_seed(5)
    v = torch.empty((n, 4), device="cuda")
    if shape == "clusters":
        c = torch.rand((8, 3), generator=g, device="cuda")
        v[:, :3] = c[torch.randint(0, 8, (n,), generator=g, device="cuda")] + 0.004 * torch.randn((n, 3), generator=g, device="cuda")
        v[:, 3] = 1e-4
    elif shape == "one_cluster":
        v[:, :3] = 0.5 + 0.001 * torch.randn((n, 3), generator=g, device="cuda")
        v[:, 3] = 1e-4
        v[0, :3] = 100.0  # a far outlier collapses the grid: all other leaves share a handful of codes
    else:
        base = torch.rand((1000, 4), generator=g, device="cuda")
        v = base[torch.randint(0, 1000, (n,), generator=g, device="cuda")].contiguous()
    b = ibvh.BVH(v)
    torch.cuda.synchronize()
    assert int(b._skew[0]) >= 1
    _, leaves = check_build_properties(v, b)
    b2 = ibvh.BVH(v, cache=b)  # (reuses the buffers; launches as many levels as the first build reported, plus one)
    torch.cuda.synchronize()
    assert b2.leaves.to_numpy().tobytes() == leaves.tobytes()


def _timed_build(v, cache):
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = ibvh.BVH(v, cache=cache)
    torch.cuda.synchronize()
    return b, (time.perf_counter() - t0) * 1e3


def test_ten_million_leaves_changing_under_a_cache_chain_time_bounds(monkeypatch):
    """the step on which a `cache=` chain's input changes.  (1) A cloud that contracts by 2 % per step — how
    inputs become clustered in a simulation — never meets a slow step, also without the always-on spare level of large
    builds: the build reports its fullest cell and the level comes on before a cell overflows.  (2) The policy of builds
    of api.SPARE_ALWAYS_FROM leaves and more (the spare level is always launched), forced here: an ABRUPT change from
    uniform to one tight cluster stays within a few normal steps.  (3) Occupancy policy alone: that one step's crowded cell is sorted by the
    finish kernel's rescue workgroups (round 6; one workgroup before: 118 ms) and the step after is normal again.  Generous bounds: the measured figures are 0.86 ms (1), 0.64 ms
    (2), 118 ms then 0.78 ms (3) against a uniform step of 0.6 ms."""
    from implicitbvh_amd import api
    n = 10_000_000
    g = torch.Generator(device="cuda").manual_seed(3)
    uniform = torch.rand((n, 4), generator=g, device="cuda") * torch.tensor([1, 1, 1, 1e-4], device="cuda")
    one = torch.empty((n, 4), device="cuda")
    one[:, :3] = 0.5 + 0.001 * torch.randn((n, 3), generator=g, device="cuda")
    one[:, 3] = 1e-4
    one[0, :3] = 100.0
    b = None
    for _ in range(4):
        b, t_uniform = _timed_build(uniform, b)
    assert int(b._skew[0]) == 0 and 0 < b._skew.occupancy() < api.SPARE_OCCUPANCY
    # (1) gradual, by the occupancy policy alone
    monkeypatch.setattr(api, "SPARE_ALWAYS_FROM", 1 << 62)
    for _ in range(2):
        b, _ = _timed_build(uniform, b)
    v = uniform.clone()
    worst = 0.0
    for step in range(40):
        v[:, :3] = 0.5 + (v[:, :3] - 0.5) * 0.98
        v[0, :3] = 0.0  # two fixed outliers keep the grid's extent: the cloud really gets denser in it
        v[1, :3] = 1.0
        b, t = _timed_build(v, b)
        if step >= 2:
            worst = max(worst, t)
    m = b.leaves.morton
    assert bool((m[1:] >= m[:-1]).all()) and int(b._skew[0]) >= 1
    assert worst < 8 * t_uniform + 2.0, (worst, t_uniform)
    # (2) abrupt, with the policy of the largest builds (SPARE_ALWAYS_FROM leaves and more): the spare level is always there
    monkeypatch.setattr(api, "SPARE_ALWAYS_FROM", 1 << 22)
    for _ in range(3):
        b, _ = _timed_build(uniform, b)
    b, t_change = _timed_build(one, b)
    m = b.leaves.morton
    assert bool((m[1:] >= m[:-1]).all())
    assert t_change < 8 * t_uniform + 2.0, (t_change, t_uniform)
    # (3) abrupt, occupancy policy alone (what smaller builds get): slow once, right, and normal on the next step
    monkeypatch.setattr(api, "SPARE_ALWAYS_FROM", 1 << 62)
    for _ in range(3):
        b, _ = _timed_build(uniform, b)
    b, t_slow = _timed_build(one, b)
    m = b.leaves.morton
    assert bool((m[1:] >= m[:-1]).all()) and int(b._skew[0]) >= 1
    # (round 6: the crowded cell is shared by the finish kernel's rescue workgroups — 118 ms with one workgroup)
    assert t_slow < 8 * t_uniform + 2.0, (t_slow, t_uniform)
    b, t_next = _timed_build(one, b)
    assert t_next < 8 * t_uniform + 2.0, (t_next, t_slow, t_uniform)
    print(f"uniform {t_uniform:.3f} ms, gradual worst {worst:.3f} ms, abrupt with spare level {t_change:.3f} ms, "
          f"abrupt without {t_slow:.3f} ms, step after {t_next:.3f} ms")


@pytest.mark.parametrize("n", [12_500_000, 13_500_000])
def test_build_properties_where_the_sort_changes_geometry(n):
    """config 5's per-GPU size (the 8,192-record finish workgroup at 75 % average fill) and the first size sorted with
    4,096 cells: the same size-independent checks as at the north-star size."""
    r0 = 0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3)
    vols = ibvh.generate_spheres(n, 7, r0=r0)
    g = ibvh.BVH(vols)
    torch.cuda.synchronize()
    assert int(g._skew[0]) == 0  # a uniform cloud crowds no cell of either geometry
    check_build_properties(vols, g)


def test_reference_published_workload_size_matches_the_oracle():
    """The reference's only published workload (README.md:226-231; benchmark/bvh_contact.jl:21-45, bvh_rays.jl:36-58): 249,882
    triangle spheres (the torus generator cut to that size; the real mesh is not in the reference repo), build, LVT
    self-traverse, traverse_rays with 100,000 rays — the size `bench.py` reports as configs.readme_250k.  Whole lists, in order,
    against the oracle."""
    import bench
    vols, _ = bench.readme_mesh_volumes(ibvh, torch)
    n = int(vols.shape[0])
    assert n == bench.README_TRIANGLES
    host = vols.cpu().numpy()
    o = orc.build(host, abi.make_types())
    g = ibvh.BVH(vols)
    assert g.leaves.to_numpy().tobytes() == o.leaves.tobytes()
    assert g.nodes.cpu().numpy().tobytes() == o.nodes.tobytes()
    for cached in (None, "again"):  # (the counting pass + write, then the enqueue path against the cached buffer)
        trav = ibvh.traverse(g, cache=None if cached is None else trav)
        exp, _ = orc.traverse_lvt(o)
        got = trav.contacts.cpu().numpy()
        assert got.shape[0] == len(exp) and (got[:, 0] == exp["a"]).all() and (got[:, 1] == exp["b"]).all()
    from implicitbvh_amd.synthetic import random_rays
    hv = host[:, :3]
    p, d = random_rays(bench.README_RAYS, hv.min(0), hv.max(0), seed=43)
    rays = ibvh.traverse_rays(g, torch.from_numpy(p).cuda().t(), torch.from_numpy(d).cuda().t())
    rexp, _ = orc.traverse_rays_lvt(o, p, d)
    rgot = rays.contacts.cpu().numpy()
    assert rgot.shape[0] == len(rexp) and (rgot[:, 0] == rexp["a"]).all() and (rgot[:, 1] == rexp["b"]).all()
    bfs = ibvh.traverse(g, ibvh.BFSTraversal())
    assert sorted(map(tuple, bfs.contacts.cpu().numpy().tolist())) == sorted(zip(exp["a"].tolist(), exp["b"].tolist()))
