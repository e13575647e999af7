"""Randomised differential testing of the HIP library against the oracle (run on a GPU box):
    python tests/fuzz_gpu.py [seconds] [seed]
Random sizes, type combinations, densities (sparse ... everything overlaps), clustered / duplicated inputs, start levels,
narrow menu, cache reuse chains, pair traversals in both orders and ray batches; every LVT list must equal the
oracle's INCLUDING order, every BFS list as a sorted set with the same num_checks.  tests/test_gpu_fuzz.py runs a
short, seeded slice of it in the suite."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root, when run as a script

import oracle_lib as orc

import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import abi

NP_F = {abi.F32: np.float32, abi.F64: np.float64}
TOKENS = {abi.BSPHERE: ibvh.BSphere, abi.BBOX: ibvh.BBox}
COMBOS = [
    (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), (abi.BSPHERE, abi.F32, abi.BSPHERE, abi.F32),
    (abi.BBOX, abi.F32, abi.BBOX, abi.F32), (abi.BSPHERE, abi.F64, abi.BBOX, abi.F32),
    (abi.BSPHERE, abi.F64, abi.BBOX, abi.F64), (abi.BSPHERE, abi.F64, abi.BSPHERE, abi.F64),
    (abi.BSPHERE, abi.F64, abi.BSPHERE, abi.F32), (abi.BBOX, abi.F64, abi.BBOX, abi.F64),
    (abi.BBOX, abi.F64, abi.BBOX, abi.F32),
]


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def pairs(c):
    return np.stack([c["a"], c["b"]], axis=1).astype(np.int64) if len(c) else np.zeros((0, 2), np.int64)


def got(t):
    return t.contacts.cpu().numpy().astype(np.int64)


def cloud(rng, n, kind, flt):
    f = NP_F[flt]
    style = rng.integers(0, 5)
    scale = float(rng.choice([0.05, 1.0, 4.0, 12.0, 40.0]))
    if style == 0:      # uniform
        c = scale * rng.random((n, 3))
    elif style == 1:    # clusters
        k = int(rng.integers(1, 8))
        c = scale * rng.random((k, 3))[rng.integers(0, k, n)] + 0.3 * rng.standard_normal((n, 3))
    elif style == 2:    # a sheet (mesh-like)
        c = scale * rng.random((n, 3))
        c[:, 2] = 0.05 * np.sin(c[:, 0])
    elif style == 3:    # many exact duplicates
        m = max(1, n // int(rng.integers(2, 6)))
        c = (scale * rng.random((m, 3)))[rng.integers(0, m, n)]
    else:               # a line
        c = np.zeros((n, 3))
        c[:, 0] = scale * rng.random(n)
    c = c.astype(f)
    size = float(rng.choice([0.02, 0.3, 1.0, 3.0]))
    if kind == abi.BSPHERE:
        r = (size * (0.1 + 0.9 * rng.random((n, 1)))).astype(f)
        return np.concatenate([c, r], axis=1)
    h = (size * (0.1 + 0.9 * rng.random((n, 3)))).astype(f)
    return np.concatenate([c - h, c + h], axis=1)


def build(vols, types, built_level=1):
    o = orc.build(vols, types, built_level=built_level)
    node_type = TOKENS[types.node_kind](torch.float32 if types.node_float == abi.F32 else torch.float64)
    opts = ibvh.BVHOptions(index=abi.INDEX_DTYPES[types.index_type],
                           morton=ibvh.DefaultMortonAlgorithm(abi.MORTON_DTYPES[types.morton_type]))
    g = ibvh.BVH(cuda(vols.astype(NP_F[types.leaf_float])), node_type, built_level=built_level, options=opts)
    return o, g


def one_case(rng, log, light=False):
    combo = COMBOS[rng.integers(0, len(COMBOS))]
    it, mt = [(abi.I32, abi.U32), (abi.I64, abi.U64), (abi.I32, abi.U16), (abi.I64, abi.U32)][rng.integers(0, 4)]
    types = abi.make_types(*combo, it, mt)
    sizes = [1, 2, 3, 7, 64, 65, 127, 129, 500, 2047, 2049, 5000, 20000, 60000]
    n = int(rng.choice(sizes[:-2] if light else sizes))  # (light: the in-suite slices keep the oracle's share small)
    vols = cloud(rng, n, combo[0], combo[1])
    levels0 = orc.tree_shape(n).levels
    built = 1 if rng.random() < 0.6 else int(rng.integers(1, levels0 + 1))  # partial builds: nodes above `built` do not exist
    o, g = build(vols, types, built)
    levels = o.tree.levels
    log.append(f"combo={combo} it={it} mt={mt} n={n} levels={levels} built={built}")
    gl = g.leaves.to_numpy()
    for field in ("morton", "index"):  # (field-wise: UInt16 codes leave two padding bytes in the record)
        assert gl[field].tolist() == o.leaves[field].tolist(), field
    assert gl["volume"].tobytes() == o.leaves["volume"].tobytes(), "leaf volumes"
    if len(o.nodes):
        lo = orc.memory_index(o.tree, 2 ** (min(built, levels - 1) - 1)) - 1 if levels > 1 else 0  # first built node
        gn = g.nodes.cpu().numpy()
        assert gn[lo:].tobytes() == o.nodes.view(gn.dtype).reshape(gn.shape)[lo:].tobytes(), "nodes"
    # LVT self, a few start levels, narrow menu, cache chain
    cache = None
    for sl in sorted({built, max(built, levels // 2), max(built, levels - 1), levels} & set(range(built, levels + 1))):
        nar = int(rng.choice([abi.NARROW_NONE, abi.NARROW_NONE, abi.NARROW_MORTON_LT, abi.NARROW_INDEX_LT]))
        exp = orc.traverse_lvt(o, sl, narrow=nar)
        if len(exp[0]) > 6_000_000:
            continue
        t = ibvh.traverse(g, start_level=sl, narrow=nar if nar else None, cache=cache if rng.random() < 0.7 else None)
        assert (got(t) == pairs(exp[0])).all(), f"lvt self sl={sl} narrow={nar}"
        if o.tree.real_nodes > 1:  # (a single-leaf tree returns empty caches, traverse_single.jl:17-21)
            assert t.cache2.cpu().numpy()[:n].tolist() == exp[1].tolist(), "inclusive counts"
        cache = t
    # BFS self
    if n <= 20000:
        sl = int(rng.integers(built, levels + 1))
        eb, res = orc.traverse_bfs(o, sl)
        if len(eb) < 3_000_000:
            b = ibvh.traverse(g, ibvh.BFSTraversal(), start_level=sl)
            assert sorted(map(tuple, got(b).tolist())) == sorted(map(tuple, pairs(eb).tolist())), "bfs self"
            assert b.num_checks == res.num_checks, "bfs checks"
    # pair, both orders
    n2 = int(rng.choice([1, 5, 64, 300, 4000, 15000]))
    other = cloud(rng, n2, combo[0], combo[1])
    o2, g2 = build(other, types)
    for (oa, ga, ob, gb) in ((o, g, o2, g2), (o2, g2, o, g)):
        sl1 = int(rng.integers(oa.built_level, oa.tree.levels + 1))
        sl2 = int(rng.integers(ob.built_level, ob.tree.levels + 1))
        nar = int(rng.choice([abi.NARROW_NONE, abi.NARROW_INDEX_LT]))
        exp = orc.traverse_pair_lvt(oa, ob, sl1, sl2, narrow=nar)[0]
        if len(exp) > 6_000_000:
            continue
        t = ibvh.traverse(ga, gb, start_level1=sl1, start_level2=sl2, narrow=nar if nar else None,
                          cache=cache if rng.random() < 0.5 else None)
        assert (got(t) == pairs(exp)).all(), f"lvt pair sl=({sl1},{sl2}) narrow={nar}"
        cache = t
    # rays (same float type for leaves and nodes only)
    if combo[1] == combo[3]:
        f = NP_F[combo[1]]
        nr = int(rng.choice([1, 63, 700, 5000]))
        lo = vols[:, :3].min(0) - 1
        hi = vols[:, :3].max(0) + 1
        p = (lo + (hi - lo) * rng.random((nr, 3))).astype(f)
        d = rng.standard_normal((nr, 3)).astype(f)
        d[rng.random(nr) < 0.1, rng.integers(0, 3)] = 0
        sl = int(rng.integers(built, levels + 1))
        exp = orc.traverse_rays_lvt(o, p, d, sl)[0]
        if len(exp) < 6_000_000:
            t = ibvh.traverse_rays(g, cuda(p).t(), cuda(d).t(), start_level=sl, cache=cache if rng.random() < 0.5 else None)
            assert (got(t) == pairs(exp)).all(), f"rays sl={sl}"


def main(seconds=60.0, seed=0, verbose=True, light=False):
    rng = np.random.default_rng(seed)
    t0, cases = time.time(), 0
    while time.time() - t0 < seconds:
        log = []
        try:
            one_case(rng, log, light)
        except Exception:
            print("FAILED case", cases, "seed", seed, *log, file=sys.stderr)
            raise
        cases += 1
    if verbose:
        print(f"fuzz ok: {cases} cases in {time.time() - t0:.1f} s (seed {seed})")
    return cases


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
         light=len(sys.argv) > 3 and sys.argv[3] == "light")
