"""Seeded random stress of every traversal against the oracle: random sizes (1 .. 6000 leaves), densities from sparse
to heavily overlapping, all nine leaf / node type combinations, both index types, random start levels, narrow on and
off, cached and fresh buffers.  LVT lists identical including order; BFS as sets with identical num_checks."""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi  # noqa: E402
from test_gpu_parity import ALL_COMBOS, NP_F, build_both, contacts_np, cuda, oracle_pairs, random_volumes  # noqa: E402


def as_set(c):
    return sorted(map(tuple, np.asarray(c).tolist()))


def check_one(seed):
    rng = np.random.default_rng(7000 + seed)
    combo = ALL_COMBOS[int(rng.integers(0, len(ALL_COMBOS)))]
    it = abi.I32 if rng.random() < 0.6 else abi.I64
    mt = [abi.U32, abi.U64, abi.U16][int(rng.integers(0, 3))]
    types = abi.make_types(*combo, it, mt)
    n = int(rng.choice([1, 2, 3, 7, 64, 65, 129, 500, 2000, 6000])) + int(rng.integers(0, 3))
    scale = float(rng.choice([1.0, 4.0, 12.0]))          # smaller box: denser cloud
    size = float(rng.choice([0.05, 0.3, 1.0]))
    vols = random_volumes(rng, n, combo[0], combo[1], scale=scale, size=size)
    o, g = build_both(vols, types)
    narrow = [None, ibvh.NARROW_MORTON_LT, ibvh.NARROW_INDEX_LT][int(rng.integers(0, 3))]
    ncode = {None: 0, ibvh.NARROW_MORTON_LT: abi.NARROW_MORTON_LT, ibvh.NARROW_INDEX_LT: abi.NARROW_INDEX_LT}[narrow]
    sl = int(rng.integers(1, o.tree.levels + 1))
    # ---- self: LVT (order), BFS (set + checks), twice through cache=
    exp = oracle_pairs(orc.traverse_lvt(o, sl, narrow=ncode)[0])
    t = ibvh.traverse(g, start_level=sl, narrow=narrow)
    assert (contacts_np(t) == exp).all() and t.num_contacts == len(exp), ("lvt", seed)
    t2 = ibvh.traverse(g, start_level=sl, narrow=narrow, cache=t)
    assert (contacts_np(t2) == exp).all(), ("lvt cache", seed)
    eb, res = orc.traverse_bfs(o, sl, narrow=ncode)
    b = ibvh.traverse(g, ibvh.BFSTraversal(), start_level=sl, narrow=narrow)
    assert as_set(contacts_np(b)) == as_set(oracle_pairs(eb)) and b.num_checks == res.num_checks, ("bfs", seed)
    b2 = ibvh.traverse(g, ibvh.BFSTraversal(), start_level=sl, narrow=narrow, cache=b)
    assert as_set(contacts_np(b2)) == as_set(oracle_pairs(eb)) and b2.num_checks == res.num_checks, ("bfs cache", seed)
    # ---- pair against a second cloud of another size
    n2 = int(rng.choice([1, 5, 64, 300, 1500])) + int(rng.integers(0, 3))
    vols2 = random_volumes(rng, n2, combo[0], combo[1], scale=scale, size=size)
    o2, g2 = build_both(vols2, types)
    s1, s2 = int(rng.integers(1, o.tree.levels + 1)), int(rng.integers(1, o2.tree.levels + 1))
    exp = oracle_pairs(orc.traverse_pair_lvt(o, o2, s1, s2, narrow=ncode)[0])
    t = ibvh.traverse(g, g2, start_level1=s1, start_level2=s2, narrow=narrow)
    assert (contacts_np(t) == exp).all(), ("pair lvt", seed)
    eb, res = orc.traverse_pair_bfs(o, o2, s1, s2, narrow=ncode)
    b = ibvh.traverse(g, g2, ibvh.BFSTraversal(), start_level1=s1, start_level2=s2, narrow=narrow)
    assert as_set(contacts_np(b)) == as_set(oracle_pairs(eb)) and b.num_checks == res.num_checks, ("pair bfs", seed)
    # ---- rays (leaf and node element types must agree)
    if combo[1] == combo[3]:
        f = NP_F[combo[1]]
        nr = int(rng.choice([1, 63, 300]))
        p = (scale * (1.2 * rng.random((nr, 3)) - 0.1)).astype(f)
        d = (rng.random((nr, 3)) - 0.5).astype(f)
        d[rng.random(nr) < 0.1, int(rng.integers(0, 3))] = 0
        rs = int(rng.integers(1, o.tree.levels + 1))
        exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d, rs)[0])
        r = ibvh.traverse_rays(g, cuda(p).t(), cuda(d).t(), start_level=rs)
        assert (contacts_np(r) == exp).all(), ("rays lvt", seed)
        eb, res = orc.traverse_rays_bfs(o, p, d, rs)
        rb = ibvh.traverse_rays(g, cuda(p).t(), cuda(d).t(), ibvh.BFSTraversal(), start_level=rs)
        assert as_set(contacts_np(rb)) == as_set(oracle_pairs(eb)) and rb.num_checks == res.num_checks, ("rays bfs", seed)


@pytest.mark.parametrize("seed", range(24))
def test_random_traversals(seed):
    check_one(seed)
