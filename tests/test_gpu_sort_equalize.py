"""Equalised cells of the build's sort (round 5; include/ibvh.h ibvh_build_desc.sort_equalize, ibvh_msd.hip splitter_kernel /
bucket_hist_kernel / partition_kernel<EQ>): cells = key ranges between splitters taken from a sorted sample instead of the cells
of a regular Morton grid.  The RESULT must not depend on the route: every build here is compared byte for byte (leaves, nodes)
with the plain route's, the small ones with the oracle as well (replaces AK.sort!, reference src/build.jl:248-253)."""
import math

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi, lib  # noqa: E402
from test_gpu_parity import NP_F, TOKENS, assert_bvh_equal, cuda, make_options  # noqa: E402


@pytest.fixture
def knob():
    yield lambda v: lib.set_tuning("msd_equalize", v)
    lib.set_tuning("msd_equalize", 0)


def clouds(n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.rand(s, generator=g, device="cuda")  # noqa: E731
    out = {}
    out["uniform"] = r(n, 3)
    c = r(8, 3)
    out["8 tight clusters"] = c[torch.randint(0, 8, (n,), generator=g, device="cuda")] + 0.004 * torch.randn((n, 3), generator=g, device="cuda")
    u, v = 2 * math.pi * r(n), 2 * math.pi * r(n)  # a surface: most cells of the grid stay empty
    out["torus surface"] = torch.stack([(1 + 0.3 * torch.cos(v)) * torch.cos(u), (1 + 0.3 * torch.cos(v)) * torch.sin(u), 0.3 * torch.sin(v)], 1)
    d = r(n, 3)
    d[: n - n // 50] = d[0]  # 98 % exact duplicates of one point
    out["duplicates"] = d[torch.randperm(n, generator=g, device="cuda")]
    few = r(1000, 3)
    out["1000 distinct centres"] = few[torch.randint(0, 1000, (n,), generator=g, device="cuda")]
    line = r(n, 1)
    out["a line"] = torch.cat([line, 0.5 * line, 0.25 + 0 * line], 1)
    out["one cluster + outlier"] = 1e-4 * torch.randn((n, 3), generator=g, device="cuda")
    out["one cluster + outlier"][n // 2] = 1000.0
    return out


@pytest.mark.parametrize("n", [4096, 5000, 77_777, 1_000_000, 3_300_000])
def test_equalised_route_is_byte_identical_to_the_plain_route(n, knob):
    for name, c in clouds(n, 5 + n % 7).items():
        for dt, mt in ((torch.float32, abi.U32), (torch.float32, abi.U64), (torch.float64, abi.U32)):
            if n > 1_000_000 and mt == abi.U64 and name not in ("8 tight clusters", "torus surface"):
                continue
            vols = torch.cat([c.to(dt), torch.full((n, 1), 1e-4, dtype=dt, device="cuda")], 1).contiguous()
            opts = ibvh.BVHOptions(morton=ibvh.DefaultMortonAlgorithm({abi.U32: np.uint32, abi.U64: np.uint64}[mt]))
            knob(-1)
            plain = ibvh.BVH(vols, options=opts)
            knob(1)
            eq = ibvh.BVH(vols, options=opts)  # (cold: two extra levels)
            for levels in (None, 0, 3):
                if levels is not None:
                    eq._skew[0] = levels  # (what the rebuild reads: 0 / 4 extra levels)
                    eq = ibvh.BVH(vols, options=opts, cache=eq)
                torch.cuda.synchronize()
                assert torch.equal(eq.leaves.buf, plain.leaves.buf), (name, n, dt, mt, levels)
                assert torch.equal(eq.nodes.view(torch.uint8), plain.nodes.view(torch.uint8)), (name, n, dt, mt, levels)
            del plain, eq


@pytest.mark.parametrize("seed", range(6))
def test_equalised_route_matches_the_oracle_on_every_record_layout(seed, knob):
    """forced on (knob = 1) for cold builds too: BSphere / BBox, F32 / F64, Int32 / Int64, UInt16 / UInt32 / UInt64 codes"""
    rng = np.random.default_rng(77 + seed)
    n = int(rng.integers(4096, 60_000))
    combos = [((abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U32), ((abi.BSPHERE, abi.F64, abi.BBOX, abi.F64), abi.I64, abi.U64),
              ((abi.BBOX, abi.F32, abi.BBOX, abi.F32), abi.I32, abi.U64), ((abi.BBOX, abi.F64, abi.BBOX, abi.F64), abi.I64, abi.U32),
              ((abi.BSPHERE, abi.F32, abi.BBOX, abi.F32), abi.I64, abi.U16), ((abi.BSPHERE, abi.F64, abi.BSPHERE, abi.F64), abi.I32, abi.U16)]
    combo, it, mt = combos[seed % len(combos)]
    f = NP_F[combo[1]]
    k = int(rng.integers(1, 6))
    c = rng.random((k, 3))[rng.integers(0, k, n)] + 10.0 ** rng.integers(-6, 0) * rng.normal(0, 1, (n, 3))
    if combo[0] == abi.BSPHERE:
        vols = np.concatenate([c, 1e-3 * rng.random((n, 1))], axis=1).astype(f)
    else:
        h = 1e-3 * rng.random((n, 3))
        vols = np.concatenate([c - h, c + h], axis=1).astype(f)
    types = abi.make_types(*combo, it, mt)
    o = orc.build(vols, types)
    node_type = TOKENS[types.node_kind](torch.float32 if types.node_float == abi.F32 else torch.float64)
    knob(1)
    g = ibvh.BVH(cuda(vols), node_type, options=make_options(types))
    assert_bvh_equal(o, g)
    # and as the in-place rebuild of pre-wrapped records (user indices kept)
    g2 = ibvh.BVH(g.leaves, node_type, options=make_options(types), cache=g)
    assert_bvh_equal(o, g2)
    torch.cuda.synchronize()
    m = g2.leaves.morton
    assert bool((m[1:] >= m[:-1]).all())


def test_a_chain_switches_to_equalised_cells_when_its_input_is_skewed_and_back():
    n = 400_000
    cl = clouds(n, 3)
    wrap = lambda c: torch.cat([c, torch.full((n, 1), 1e-4, device="cuda")], 1).contiguous()  # noqa: E731
    uni, tight = wrap(cl["uniform"]), wrap(cl["8 tight clusters"])
    ref_u, ref_t = ibvh.BVH(uni), ibvh.BVH(tight)
    b = ibvh.BVH(uni)
    torch.cuda.synchronize()  # (a rebuild reads the hint WITHOUT waiting for it: what a cold build starts with asks for extra levels)
    asked = []
    for step, v in enumerate([uni, uni, tight, tight, tight, tight, uni, uni, uni]):
        b = ibvh.BVH(v, cache=b)
        torch.cuda.synchronize()
        asked.append(int(b._fast[1].sort_equalize))
        ref = ref_u if v is uni else ref_t
        assert torch.equal(b.leaves.buf, ref.leaves.buf) and torch.equal(b.nodes, ref.nodes), step
    # uniform steps never ask; the first clustered step is met by the plain grid (its hint then asks), the following ones are
    # equalised and stay so (bit 16 of the hint), the first uniform step after them still is, then the chain falls back
    assert asked == [0, 0, 0, 1, 1, 1, 1, 0, 0], asked


def test_a_chain_of_duplicates_goes_back_to_the_plain_grid_for_a_while():
    """One tight cluster and a far outlier (every leaf shares one of 1 - 8 Morton codes): no choice of cells splits a run of equal
    keys, the equalised build reports that most records sat in crowded cells all the same (hint bit 17) and the chain stays with
    the plain grid for api.EQ_HOLDOFF rebuilds before it tries again."""
    from implicitbvh_amd import api
    n = 1_000_000
    v = torch.cat([clouds(n, 4)["one cluster + outlier"], torch.full((n, 1), 1e-4, device="cuda")], 1).contiguous()
    ref = ibvh.BVH(v)
    b = ibvh.BVH(v)
    torch.cuda.synchronize()
    asked = []
    for step in range(api.EQ_HOLDOFF + 2):
        b = ibvh.BVH(v, cache=b)
        torch.cuda.synchronize()
        asked.append(int(b._fast[1].sort_equalize))
        assert torch.equal(b.leaves.buf, ref.leaves.buf), step
    assert asked == [1] + [0] * api.EQ_HOLDOFF + [1], asked


def test_equalised_route_with_4096_cells_of_64_bit_codes(knob):
    """1.5e7 leaves take 4,096 first-level cells; with UInt64 codes the histogram kernel's splitters, search tree and counters are
    80 KB of LDS (beyond the default limit of a launch): byte-identical to the plain route."""
    n = 15_000_000
    c = clouds(n, 11)["8 tight clusters"]
    vols = torch.cat([c, torch.full((n, 1), 1e-4, device="cuda")], 1).contiguous()
    del c
    opts = ibvh.BVHOptions(morton=ibvh.DefaultMortonAlgorithm(np.uint64))
    knob(-1)
    plain = ibvh.BVH(vols, options=opts)
    knob(1)
    eq = ibvh.BVH(vols, options=opts)
    torch.cuda.synchronize()
    assert torch.equal(eq.leaves.buf, plain.leaves.buf) and torch.equal(eq.nodes, plain.nodes)


@pytest.mark.parametrize("shape", ["8 tight clusters", "torus surface", "duplicates"])
def test_equalised_route_on_other_record_layouts_at_1e6(shape, knob):
    """BBox leaves (40-byte records), Int64 indices with UInt16 codes (30 leaves per code: every cell is runs of equal keys), and the
    in-place rebuild of pre-wrapped records: byte-identical to the plain route."""
    n = 1_000_000
    c = clouds(n, 21)[shape]
    h = torch.full((n, 3), 1e-4, device="cuda")
    cases = [("BBox{Float32} leaves", torch.cat([c - h, c + h], 1).contiguous(), ibvh.BVHOptions()),
             ("Int64 indices, UInt16 codes", torch.cat([c, h[:, :1]], 1).contiguous(),
              ibvh.BVHOptions(index=np.int64, morton=ibvh.DefaultMortonAlgorithm(np.uint16)))]
    for what, vols, opts in cases:
        knob(-1)
        plain = ibvh.BVH(vols, options=opts)
        knob(1)
        eq = ibvh.BVH(vols, options=opts)
        torch.cuda.synchronize()
        assert torch.equal(eq.leaves.buf, plain.leaves.buf) and torch.equal(eq.nodes, plain.nodes), (what, shape)
        # the in-place rebuild of the records (user indices kept), both routes
        ref = plain.leaves.buf.clone()
        for route in (-1, 1):
            knob(route)
            b = ibvh.BVH(eq.leaves, options=opts, cache=eq)
            torch.cuda.synchronize()
            assert torch.equal(b.leaves.buf, ref) and torch.equal(b.nodes, plain.nodes), (what, shape, route)
            eq = b
