"""Parity tests of the development VARIANTS (variants/*.inc) — not part of the product suite: run ON THE GPU BOX against a
library built with `tools/build_variant.sh dev -DIBVH_VARIANTS` (full types: drop -DIBVH_ONLY_BENCH_TYPES there), e.g.
    IBVH_LIB=variants/libibvh_dev.so python -m pytest variants/test_variants_gpu.py -q
The kernels behind the knobs "rays_shadow" and "lvt_dual" are not in libibvh.so."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as orc  # noqa: E402
from test_gpu_parity import (_positions, _rays_positions, build_both, contacts_np, cuda, oracle_pairs, random_volumes)  # noqa: E402
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi, lib  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", [abi.BSPHERE, abi.BBOX], ids=["spheres", "boxes"])
def test_rays_over_the_wide_shadow_identical_order(kind):
    """With the development knob "rays_shadow" single-precision trees with BBox nodes walk a quantised 8-wide shadow of the
    node levels (ibvh_rays_scratch_bytes, csrc/ibvh_lvt.hip "(3b)"); regular rays go there, irregular ones (zero / infinite / NaN components: the slab test is
    not monotone for them) to the binary walker behind it.  Every mix must give the oracle's list in the oracle's order:
    uncached, through the cached (enqueue) path, with a tiny contact cache (blocks that overflow and walk again), with
    the ray narrow and with positions; trees of several heights incl. ragged last levels."""
    from implicitbvh_amd import api
    lib.set_tuning("rays_shadow", 1)   # (development knob: the shadow walker is off by default — slower on config 3)
    api._shape_memo.clear()            # (scratch sizes depend on the knob)
    try:
        _rays_over_the_wide_shadow(kind)
    finally:
        lib.set_tuning("rays_shadow", 0)
        api._shape_memo.clear()


def _rays_over_the_wide_shadow(kind):
    rng = np.random.default_rng(77)
    types = abi.make_types(kind, abi.F32, abi.BBOX, abi.F32)
    for n in (129, 1000, 4097, 40_001):
        vols = random_volumes(rng, n, kind, abi.F32, scale=20.0 if n > 2000 else 8.0)
        o, g = build_both(vols, types)
        assert o.tree.levels >= 9
        nr = 3000
        p = (rng.random((nr, 3)) * (22 if n > 2000 else 9) - 1).astype(np.float32)
        d = (rng.random((nr, 3)) - 0.5).astype(np.float32)
        d[::5, rng.integers(0, 3)] = 0            # zero components: irregular
        d[7::41] = 0                               # the null direction
        d[3::53, 1] = np.inf
        d[11::67, 2] = 1e-45                       # 1/d overflows: irregular
        p[13::71, 0] = np.nan
        p[17::73, 1] = np.inf
        d[19::79] *= 1e30                          # huge but finite: regular
        with np.errstate(all="ignore"):
            exp = oracle_pairs(orc.traverse_rays_lvt(o, p, d)[0]).reshape(-1, 2)
        P_, D_ = cuda(p).t(), cuda(d).t()
        t1 = ibvh.traverse_rays(g, P_, D_)
        assert (contacts_np(t1).reshape(-1, 2) == exp).all()
        t2 = ibvh.traverse_rays(g, P_, D_, cache=t1)   # enqueue path against the cached buffer
        assert (contacts_np(t2).reshape(-1, 2) == exp).all()
        from implicitbvh_amd import api
        keep = api.RAY_CACHE_SLOTS
        try:
            api.RAY_CACHE_SLOTS = 1                     # most blocks overflow their cache and walk again when writing
            t3 = ibvh.traverse_rays(g, P_, D_)
            assert (contacts_np(t3).reshape(-1, 2) == exp).all()
        finally:
            api.RAY_CACHE_SLOTS = keep
        # fewer rays than the shadow pays for: the binary walk, same list
        few = 1 + n // 200
        assert (contacts_np(ibvh.traverse_rays(g, P_[:, :few].contiguous(), D_[:, :few].contiguous())).reshape(-1, 2)
                == exp[exp[:, 1] <= few]).all()
        # positions + the ray narrow through the shadow walker
        pos = _positions(o.leaves)
        tp = contacts_np(ibvh.api.traverse_rays(g, P_, D_, narrow=lambda bv, pp, dd: bv.index > 0)).reshape(-1, 2)
        assert (tp == exp).all()
        raw = contacts_np(_rays_positions(g, P_, D_)).reshape(-1, 2)
        assert (raw[:, 0] == pos[exp[:, 0]]).all() and (raw[:, 1] == exp[:, 1]).all()
        bfs = ibvh.traverse_rays(g, P_, D_, ibvh.BFSTraversal())
        assert sorted(map(tuple, contacts_np(bfs).tolist())) == sorted(map(tuple, exp.tolist()))


