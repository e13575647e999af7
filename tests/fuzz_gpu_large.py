"""Randomised differential testing at the sizes tests/fuzz_gpu.py does not reach (run on a GPU box):
    python tests/fuzz_gpu_large.py [seconds] [seed]
1.3e5 .. 2.5e6 leaves, so that the paths chosen by size are the ones under test: the MSD sort with its spare levels, equalised
cells and rescue workgroups, `cache=` chains whose input changes ABRUPTLY between two builds (the hint is one build old), the
shared descent per block of the LVT count pass, the one-kernel scans, the binned ray path with its tail as units.  Every build
is compared with the oracle's byte for byte (order, codes, node volumes), every LVT list INCLUDING its order and its inclusive
counts, every ray list including order.  tests/test_gpu_fuzz.py runs one short seeded slice of it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root, when run as a script

import oracle_lib as orc

import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import abi
from implicitbvh_amd.synthetic import sphere_radius_law, torus_mesh

NP_F = {abi.F32: np.float32, abi.F64: np.float64}
TOKENS = {abi.BSPHERE: ibvh.BSphere, abi.BBOX: ibvh.BBox}
# (leaf kind, leaf float, node kind, node float, index, morton): the bench types most of the time
TYPES = [(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32, abi.I32, abi.U32)] * 4 + [
    (abi.BSPHERE, abi.F64, abi.BBOX, abi.F64, abi.I32, abi.U32), (abi.BSPHERE, abi.F64, abi.BBOX, abi.F32, abi.I64, abi.U64),
    (abi.BBOX, abi.F32, abi.BBOX, abi.F32, abi.I32, abi.U64), (abi.BSPHERE, abi.F32, abi.BSPHERE, abi.F32, abi.I32, abi.U32),
    (abi.BBOX, abi.F64, abi.BBOX, abi.F64, abi.I64, abi.U32)]
STYLES = ["uniform", "uniform", "clusters", "sheet", "torus", "one_cell", "all_equal", "duplicates", "half_collapsed", "line"]
SPARSE = {"uniform", "clusters", "sheet", "torus"}  # styles whose contact lists stay near n with the radius law


DONE = {"builds": 0, "lvt_self_lists": 0, "lvt_pair_lists": 0, "ray_lists": 0, "overflows": 0}  # what was compared so far


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def pairs(c):
    return np.stack([c["a"], c["b"]], axis=1).astype(np.int64) if len(c) else np.zeros((0, 2), np.int64)


def centres(rng, style, n):
    if style == "uniform":
        return rng.random((n, 3))
    if style == "clusters":
        k = int(rng.integers(2, 9))
        return rng.random((k, 3))[rng.choice(k, n, p=rng.dirichlet(np.ones(k)))] + 0.02 * rng.standard_normal((n, 3))
    if style == "sheet":
        c = rng.random((n, 3))
        c[:, 2] = 0.05 * np.sin(6 * c[:, 0]) * np.cos(5 * c[:, 1])
        return c
    if style == "torus":
        u = int(np.sqrt(n / 2)) + 2
        tris = torus_mesh(u, u)[:n]
        c = tris.reshape(len(tris), 3, 3).mean(1)
        return np.concatenate([c, c[: n - len(c)]]) if len(c) < n else c
    if style == "one_cell":       # nearly everything in ONE cell of the coarse grid
        c = 0.5 + 1e-3 * rng.standard_normal((n, 3))
        c[0] = 50.0
        return c
    if style == "all_equal":      # every code equal
        return np.full((n, 3), 0.25) + np.array([[1e-9, 0, 0]]) * rng.integers(0, 2, (n, 1))
    if style == "duplicates":
        m = max(1, n // int(rng.integers(3, 200)))
        return rng.random((m, 3))[rng.integers(0, m, n)]
    if style == "half_collapsed":
        c = rng.random((n, 3))
        c[rng.random(n) < 0.5] = rng.random(3)
        return c
    if style == "line":
        c = np.zeros((n, 3))
        c[:, int(rng.integers(0, 3))] = rng.random(n)
        return c
    raise ValueError(style)


def volumes(rng, style, n, kind, flt):
    c = centres(rng, style, n)
    r0 = sphere_radius_law(n) * float(rng.choice([0.5, 1.0, 1.5])) if style in SPARSE else 1e-6
    if style == "torus":
        r0 *= 3.0
    if style == "clusters":  # (a few thousand times the uniform density inside a cluster)
        r0 *= 0.06
    if kind == abi.BSPHERE:
        v = np.concatenate([c, r0 * (0.5 + 0.5 * rng.random((n, 1)))], axis=1)
    else:
        h = r0 * (0.5 + 0.5 * rng.random((n, 3)))
        v = np.concatenate([c - h, c + h], axis=1)
    return v.astype(NP_F[flt])


def check_build(o, g, what):
    gl = g.leaves.to_numpy()
    for field in ("morton", "index"):
        assert np.array_equal(gl[field], o.leaves[field]), f"{what}: {field}"
    assert gl["volume"].tobytes() == o.leaves["volume"].tobytes(), f"{what}: leaf volumes"
    gn = g.nodes.cpu().numpy()
    assert gn.tobytes() == o.nodes.view(gn.dtype).reshape(gn.shape).tobytes(), f"{what}: nodes"


def one_case(rng, log, sizes):
    t = TYPES[rng.integers(0, len(TYPES))]
    types = abi.make_types(*t)
    node_type = TOKENS[t[2]](torch.float32 if t[3] == abi.F32 else torch.float64)
    opts = ibvh.BVHOptions(index=abi.INDEX_DTYPES[t[4]], morton=ibvh.DefaultMortonAlgorithm(abi.MORTON_DTYPES[t[5]]))
    n = int(rng.choice(sizes)) + int(rng.integers(-3, 4))
    chain = [STYLES[rng.integers(0, len(STYLES))] for _ in range(int(rng.integers(2, 5)))]
    log.append(f"types={t} n={n} chain={chain}")
    g = trav = rays_cache = None
    for step, style in enumerate(chain):
        vols = volumes(rng, style, n, t[0], t[1])
        o = orc.build(vols, types)
        # `cache=` chain: the buffers AND the sort's hint come from the build before (another distribution, often)
        g = ibvh.BVH(cuda(vols), node_type, options=opts, cache=g if rng.random() < 0.85 else None)
        check_build(o, g, f"build {step} ({style})")
        DONE["builds"] += 1
        if style not in SPARSE:
            continue
        if rng.random() < 0.7:  # LVT self: list with order, inclusive counts
            try:
                trav = ibvh.traverse(g, cache=trav if rng.random() < 0.7 else None)
            except OverflowError:  # (more contacts than the index type counts: the documented error, as the reference's)
                assert t[4] == abi.I32
                trav = None
                DONE["overflows"] += 1
            if trav is not None and trav.num_contacts <= 30_000_000:
                exp = orc.traverse_lvt(o, None)
                assert np.array_equal(trav.contacts.cpu().numpy().astype(np.int64), pairs(exp[0])), f"lvt self, step {step} ({style})"
                assert np.array_equal(trav.cache2.cpu().numpy()[:n].astype(np.int64), exp[1].astype(np.int64)), "inclusive counts"
                DONE["lvt_self_lists"] += 1
        if rng.random() < 0.35:  # pair against a smaller, shifted cloud of another style
            n2 = int(rng.choice([70_000, 140_000, 400_000]))
            style2 = [s for s in STYLES if s in SPARSE][rng.integers(0, 5)]
            v2 = volumes(rng, style2, n2, t[0], t[1])
            shift = (0.3 * rng.random(3)).astype(v2.dtype)
            v2[:, :3] += shift
            if t[0] == abi.BBOX:
                v2[:, 3:] += shift
            o2 = orc.build(v2, types)
            g2 = ibvh.BVH(cuda(v2), node_type, options=opts)
            check_build(o2, g2, f"build other ({style2})")
            for (oa, ga, ob, gb) in ((o, g, o2, g2), (o2, g2, o, g)):
                try:
                    tp = ibvh.traverse(ga, gb)
                except OverflowError:
                    assert t[4] == abi.I32
                    continue
                if tp.num_contacts <= 30_000_000:
                    exp = orc.traverse_pair_lvt(oa, ob, None, None)[0]
                    assert np.array_equal(tp.contacts.cpu().numpy().astype(np.int64), pairs(exp)), f"lvt pair, step {step}"
                    DONE["lvt_pair_lists"] += 1
        if t[1] == t[3] and rng.random() < 0.5:  # rays (one float type): the binned path from ~1e5 rays on, the walker below
            nr = int(rng.choice([5_000, 120_000, 300_000]))
            f = NP_F[t[1]]
            lo, hi = vols[:, :3].min(0) - 0.05, vols[:, :3].max(0) + 0.05
            p = (lo + (hi - lo) * rng.random((nr, 3))).astype(f)
            d = rng.standard_normal((nr, 3)).astype(f)
            d[rng.random(nr) < 0.05, rng.integers(0, 3)] = 0
            tr = ibvh.traverse_rays(g, cuda(p).t(), cuda(d).t(), cache=rays_cache if rng.random() < 0.5 else None)
            if tr.num_contacts <= 20_000_000:
                exp = orc.traverse_rays_lvt(o, p, d, 1)[0]
                assert np.array_equal(tr.contacts.cpu().numpy().astype(np.int64), pairs(exp)), f"rays, step {step} ({style}, {nr} rays)"
                DONE["ray_lists"] += 1
            rays_cache = tr


def main(seconds=120.0, seed=0, verbose=True, sizes=(131_072, 200_000, 524_289, 1_000_000, 2_500_000)):
    rng = np.random.default_rng(seed)
    t0, cases = time.time(), 0
    while time.time() - t0 < seconds:
        log = []
        try:
            one_case(rng, log, sizes)
        except Exception:
            print("FAILED case", cases, "seed", seed, *log, file=sys.stderr)
            raise
        cases += 1
        if verbose:
            print(f"  case {cases} ok at {time.time() - t0:.0f} s: {log[0]}", flush=True)
    if verbose:
        print(f"fuzz ok: {cases} cases in {time.time() - t0:.1f} s (seed {seed}); compared: {DONE}")
    return cases


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
