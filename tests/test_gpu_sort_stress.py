"""Seeded random stress of the build's sort (MSD partition + extra partition levels + in-LDS finish) against the oracle:
random sizes, random mixtures of cluster scales (from a uniform cloud down to exact duplicates), 32- and 64-bit codes,
every depth of extra levels, 24- to 56-byte records.  Bit-exact leaves, nodes, skips and extrema."""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi  # noqa: E402
from test_gpu_parity import NP_F, TOKENS, assert_bvh_equal, cuda, make_options  # noqa: E402


def random_cloud(rng, n):
    """centres = a few cluster centres + noise of wildly different scales; some exact duplicates; some far outliers"""
    k = int(rng.integers(1, 9))
    centres = rng.random((k, 3)) * 10.0 ** rng.integers(-1, 3)
    scales = 10.0 ** rng.integers(-7, 1, size=k).astype(np.float64)
    scales[rng.random(k) < 0.2] = 0.0
    which = rng.choice(k, n, p=rng.dirichlet(np.ones(k) * 0.5))
    c = centres[which] + scales[which][:, None] * rng.normal(0, 1, (n, 3))
    if rng.random() < 0.3:
        c[rng.integers(0, n, size=int(rng.integers(1, 4)))] += 10.0 ** rng.integers(1, 5)
    if rng.random() < 0.3:  # runs of exact duplicates
        src = rng.integers(0, n, n // 4)
        c[rng.integers(0, n, n // 4)] = c[src]
    return c


@pytest.mark.parametrize("seed", range(24))
def test_random_clouds_every_level_depth(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(4096, 220_000))
    c = random_cloud(rng, n)
    combos = [((abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U32),
              ((abi.BSPHERE, abi.F64, abi.BBOX, abi.F64), abi.I64, abi.U64),
              ((abi.BBOX, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U64),
              ((abi.BBOX, abi.F64, abi.BBOX, abi.F64), abi.I64, abi.U32)]
    combo, it, mt = combos[seed % len(combos)]
    f = NP_F[combo[1]]
    if combo[0] == abi.BSPHERE:
        vols = np.concatenate([c, 1e-3 * rng.random((n, 1))], axis=1).astype(f)
    else:
        h = 1e-3 * rng.random((n, 3))
        vols = np.concatenate([c - h, c + h], axis=1).astype(f)
    types = abi.make_types(*combo, it, mt)
    o = orc.build(vols, types)
    node_type = TOKENS[types.node_kind](torch.float32 if types.node_float == abi.F32 else torch.float64)
    dev = cuda(vols)
    g = ibvh.BVH(dev, node_type, options=make_options(types))
    assert_bvh_equal(o, g)
    torch.cuda.synchronize()
    # the hint word a rebuild reads: low byte = extra levels (launches 0, 2, 3, 4 of them), bit 16 = equalised cells (round 5:
    # cells = key ranges between splitters taken from a sorted sample; a hint with extra levels asks for them as well unless
    # api.EQUALIZE is off — both routes, every depth)
    try:
        for equalize in (True, False):
            ibvh.api.EQUALIZE = equalize
            for pretend in (0, 1, 2, 3):
                g._skew[0] = pretend | (1 << 16 if equalize else 0)
                g = ibvh.BVH(dev, node_type, options=make_options(types), cache=g)
                assert_bvh_equal(o, g)
                torch.cuda.synchronize()
                assert 0 <= int(g._skew[0]) <= abi.MAX_SORT_LEVELS
                assert equalize or not g._skew.equalize()
    finally:
        ibvh.api.EQUALIZE = True
