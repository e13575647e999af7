"""The finish kernel's rescue workgroups (csrc/ibvh_msd_finish.hip, round 6): a range with more records than one workgroup sorts
in LDS that no partition level is left to split — an ABRUPT change of distribution inside a `cache=` chain, whose hint is one
build old — is sorted window by window and merged by all rescue workgroups instead of by one workgroup alone (118 ms at 1e7
leaves, 17 ms at 1e6 in rounds 2 - 5).  Bit-exact against the oracle for every launched depth, both key widths, both routes
(plain grid / equalised cells), the old path (knob msd_rescue = 0) beside it; and the time bound."""
import time

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi, api, lib  # noqa: E402
from test_gpu_parity import NP_F, TOKENS, assert_bvh_equal, cuda, make_options  # noqa: E402


def clouds(kind, n, rng):
    if kind == "one_cluster":            # nearly every leaf in ONE cell of the coarse grid: one range of ~n records
        c = 0.5 + 1e-3 * rng.normal(0, 1, (n, 3))
        c[0] = 100.0
    elif kind == "wide_cluster":         # one crowded cell whose keys differ in ~18 bits: the merge's rank searches at full depth
        c = 0.37 + 0.01 * rng.normal(0, 1, (n, 3))
        c[0], c[1] = 0.0, 1.0
    elif kind == "eight_clusters":       # several crowded cells of different sizes
        k = rng.random((8, 3))
        c = k[rng.choice(8, n, p=rng.dirichlet(np.ones(8)))] + 2e-3 * rng.normal(0, 1, (n, 3))
    elif kind == "duplicates":           # runs of equal keys inside the crowded range: ties keep source order (stability)
        base = 0.5 + 1e-3 * rng.normal(0, 1, (max(n // 50, 1), 3))
        c = base[rng.integers(0, len(base), n)]
        c[0] = 100.0
    elif kind == "sorted_cluster":       # already in Morton order: every output chunk comes from ONE run
        c = 0.5 + 1e-3 * np.sort(rng.random(n))[:, None] * np.ones((1, 3))
        c[-1] = 100.0
    else:
        raise ValueError(kind)
    return c


def volumes(c, combo, rng):
    f = NP_F[combo[1]]
    n = len(c)
    if combo[0] == abi.BSPHERE:
        return np.concatenate([c, 1e-4 * rng.random((n, 1))], axis=1).astype(f)
    h = 1e-4 * rng.random((n, 3))
    return np.concatenate([c - h, c + h], axis=1).astype(f)


CASES = [("one_cluster", 300_000, (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U32),
         ("one_cluster", 1_000_000, (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U32),
         ("one_cluster", 150_001, (abi.BSPHERE, abi.F64, abi.BBOX, abi.F64), abi.I64, abi.U64),
         ("wide_cluster", 600_000, (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U32),
         ("wide_cluster", 100_000, (abi.BSPHERE, abi.F64, abi.BBOX, abi.F64), abi.I64, abi.U64),
         ("eight_clusters", 500_000, (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U32),
         ("eight_clusters", 90_000, (abi.BBOX, abi.F64, abi.BBOX, abi.F64), abi.I64, abi.U32),
         ("duplicates", 400_000, (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U32),
         ("duplicates", 120_000, (abi.BBOX, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U64),
         ("sorted_cluster", 200_000, (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U32),
         ("one_cluster", 4_097, (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U32),
         ("one_cluster", 40_000, (abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U16)]


@pytest.mark.parametrize("kind,n,combo,it,mt", CASES)
def test_too_few_levels_for_the_input_every_depth_both_routes(kind, n, combo, it, mt):
    rng = np.random.default_rng(n)
    vols = volumes(clouds(kind, n, rng), combo, rng)
    types = abi.make_types(*combo, it, mt)
    o = orc.build(vols, types)
    node_type = TOKENS[types.node_kind](torch.float32 if types.node_float == abi.F32 else torch.float64)
    dev = cuda(vols)
    g = ibvh.BVH(dev, node_type, options=make_options(types))
    assert_bvh_equal(o, g)
    torch.cuda.synchronize()
    try:
        for rescue in (1, 0):
            lib.set_tuning("msd_rescue", rescue)
            for equalize in (False, True):
                api.EQUALIZE = equalize
                eq = 1 << 16 if equalize else 0
                # hint words a chain could hold -> 0, 1, 2, 3 extra levels launched (api.sort_hint_rule), whatever the input needs
                for word, levels in ((eq, 0), (125 << 8 | eq, 1), (1 | eq, 2), (2 | eq, 3)):
                    assert api.sort_hint_rule(word, n, 0)[:2] == (levels, 1 if equalize else 0)
                    g._skew[0] = word
                    g = ibvh.BVH(dev, node_type, options=make_options(types), cache=g)
                    assert_bvh_equal(o, g)
                    torch.cuda.synchronize()
    finally:
        api.EQUALIZE = True
        lib.set_tuning("msd_rescue", 1)


def _timed(v, cache):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = ibvh.BVH(v, cache=cache)
    torch.cuda.synchronize()
    return b, (time.perf_counter() - t0) * 1e3


@pytest.mark.parametrize("sigma", [0.001, 0.01])
@pytest.mark.parametrize("n", [1_000_000, 2_500_000])
def test_abrupt_change_under_the_default_policy_stays_within_a_few_steps(n, sigma):
    """uniform -> one tight cluster inside a `cache=` chain, default policy (no spare level below 2^24 leaves):
    the step of the change is bounded like every other step (17 ms at 1e6 before), the result is sorted, the hint is left."""
    g = torch.Generator(device="cuda").manual_seed(5)
    uniform = torch.rand((n, 4), generator=g, device="cuda") * torch.tensor([1, 1, 1, 1e-4], device="cuda")
    one = torch.empty((n, 4), device="cuda")
    # sigma 0.001 in a box of 100: every leaf of the cluster has the same code; 0.01 in the unit box: ~18 bits of the code vary
    one[:, :3] = (0.5 if sigma < 0.005 else 0.37) + sigma * torch.randn((n, 3), generator=g, device="cuda")
    one[:, 3] = 1e-4
    one[0, :3] = 100.0 if sigma < 0.005 else 1.0
    one[1, :3] = one[1, :3] if sigma < 0.005 else 0.0
    b = None
    for _ in range(4):
        b, t_uniform = _timed(uniform, b)
    assert int(b._skew[0]) == 0
    b, t_slow = _timed(one, b)
    m = b.leaves.morton
    assert bool((m[1:] >= m[:-1]).all()) and int(b._skew[0]) >= 1
    idx = b.leaves.index.long()
    assert int(idx.min()) == 1 and int(idx.max()) == n and int(torch.unique(idx).numel()) == n
    assert t_slow < 8 * t_uniform + 2.0, (t_slow, t_uniform)
    b, t_next = _timed(one, b)
    assert t_next < 8 * t_uniform + 2.0, (t_next, t_uniform)
    print(f"n={n} sigma={sigma}: uniform {t_uniform:.3f} ms, step of the change {t_slow:.3f} ms, step after {t_next:.3f} ms")
