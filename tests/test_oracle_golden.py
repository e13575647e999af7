"""Pins the CPU oracle against every known answer the reference's own tests / doctests hold for
the hot path.  Fixtures: tests/golden/reference_known_answers.json (data
transcribed from the reference, each entry cites its file:line)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as orc
from implicitbvh_amd import abi

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))

FLT = {"F32": abi.F32, "F64": abi.F64}
IDX = {"I32": abi.I32, "I64": abi.I64}
MOR = {"U16": abi.U16, "U32": abi.U32, "U64": abi.U64}


@pytest.mark.parametrize("case", G["tree_shapes"], ids=lambda c: f"n{c['n']}")
def test_tree_shapes(case):
    t = orc.tree_shape(case["n"])
    assert t.astuple() == (case["levels"], case["real_leaves"], case["real_nodes"], case["virtual_leaves"],
                           case["virtual_nodes"])
    for idx, mem in case["memory_index"]:
        assert orc.memory_index(t, idx) == mem
    for level, a, b in case["level_indices"]:
        assert orc.level_indices(t, level) == (a, b)
    for idx, v in case["isvirtual"]:
        assert orc.isvirtual(t, idx) == v


def test_tree_domain_error():
    with pytest.raises(abi.DomainError):
        orc.tree_shape(0)


def test_morton_split3():
    for bits in (16, 32, 64):
        assert orc.morton_split3(G["morton_split3"]["input"], bits) == G["morton_split3"]["output"]


def test_morton_kat_0x06186186():
    k = G["morton_kat"]
    types = abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32, abi.I32, abi.U32)
    bvh = orc.build(k["spheres"], types, indices=k["indices"])
    l1 = bvh.leaves[0]
    assert l1["volume"]["x"].tolist() == k["leaves1"]["x"] and float(l1["volume"]["r"]) == k["leaves1"]["r"]
    assert int(l1["index"]) == k["leaves1"]["index"]
    assert int(l1["morton"]) == int(k["leaves1"]["morton"], 16)
    # the surveyor's hand-derived codes for the other four leaves (not asserted by the reference)
    assert [int(m) for m in bvh.leaves["morton"]] == [int(v, 16) for v in k["derived_mortons_sorted"]["values"]]
    # the floatmin max-init quirk (morton/utils.jl:39-40): x/y maxima are ~2*floatmin, not 0
    fm = np.finfo(np.float32).tiny
    assert bvh.extrema[3] > 0 and bvh.extrema[3] < 4 * fm and bvh.extrema[0] == -fm


@pytest.mark.parametrize("variant", G["readme_example"]["variants"], ids=lambda v: "-".join(map(str, v.values())))
def test_readme_example_contacts_in_order(variant):
    e = G["readme_example"]
    types = abi.make_types(abi.BSPHERE, FLT[variant["leaf_float"]], abi.BBOX, abi.F32, IDX[variant["index"]],
                           MOR[variant["morton"]])
    bvh = orc.build(e["spheres"], types)
    contacts, _ = orc.traverse_lvt(bvh)
    assert orc.pairs_as_tuples(contacts) == [tuple(c) for c in e["contacts"]]
    bfs, res = orc.traverse_bfs(bvh)
    assert sorted(orc.pairs_as_tuples(bfs)) == sorted(tuple(c) for c in e["contacts"])


def test_pair_example():
    e = G["pair_example"]
    types = abi.make_types()
    b1, b2 = orc.build(e["spheres1"], types), orc.build(e["spheres2"], types)
    contacts, _ = orc.traverse_pair_lvt(b1, b2, e["start_level1"], e["start_level2"])
    assert orc.pairs_as_tuples(contacts) == [tuple(c) for c in e["contacts"]]
    bfs, _ = orc.traverse_pair_bfs(b1, b2, e["start_level1"], e["start_level2"])
    assert sorted(orc.pairs_as_tuples(bfs)) == sorted(tuple(c) for c in e["contacts"])


def test_ray_example():
    e = G["ray_example"]
    bvh = orc.build(e["spheres"], abi.make_types())
    contacts, _ = orc.traverse_rays_lvt(bvh, e["points"], e["directions"])
    assert orc.pairs_as_tuples(contacts) == [tuple(c) for c in e["contacts"]]
    bfs, _ = orc.traverse_rays_bfs(bvh, e["points"], e["directions"])
    assert sorted(orc.pairs_as_tuples(bfs)) == sorted(tuple(c) for c in e["contacts"])


@pytest.mark.parametrize("case", G["build_structure"], ids=["ordered", "unordered"])
@pytest.mark.parametrize("node_kind", [abi.BSPHERE, abi.BBOX])
def test_build_structure(case, node_kind):
    """runtests.jl:596-834: which leaves pair into which node, for sphere and box nodes (F64)."""
    types = abi.make_types(abi.BSPHERE, abi.F64, node_kind, abi.F64)
    bvh = orc.build(case["spheres"], types)
    assert len(bvh.nodes) == case["num_nodes"]
    assert bvh.leaves["index"].tolist() == case["sorted_indices"]
    sph = np.asarray(case["spheres"], np.float64)

    def vol(i):
        return sph[i - 1]

    def centre(v):
        return v["x"] if node_kind == abi.BSPHERE else 0.5 * (v["lo"] + v["up"])

    n4 = orc.merge(types, vol(case["node4"][0]), vol(case["node4"][1]))
    n5 = orc.merge(types, vol(case["node5"][0]), vol(case["node5"][1]))
    n6 = orc.merge(types, vol(case["node6"][0]))
    # nodes are 1-based in the reference: nodes[4], nodes[5], nodes[6]
    assert np.allclose(centre(bvh.nodes[3]), centre(n4))
    assert np.allclose(centre(bvh.nodes[4]), centre(n5))
    assert np.allclose(centre(bvh.nodes[5]), centre(n6))
    assert np.allclose(centre(bvh.nodes[2]), centre(n6))  # level 2, node 3 = lone child copy
    for trav in (orc.traverse_lvt, orc.traverse_bfs):
        c = trav(bvh)[0]
        assert sorted(orc.pairs_as_tuples(c)) == sorted(tuple(x) for x in case["contacts_set"])
    # BBox leaves variant (runtests.jl:655-713, 791-834): leaves = BBox(BSphere)
    boxes = np.concatenate([sph[:, :3] - sph[:, 3:4], sph[:, :3] + sph[:, 3:4]], axis=1)
    tb = abi.make_types(abi.BBOX, abi.F64, abi.BBOX, abi.F64)
    bb = orc.build(boxes, tb)
    assert bb.leaves["index"].tolist() == case["sorted_indices"]
    for trav in (orc.traverse_lvt, orc.traverse_bfs):
        assert sorted(orc.pairs_as_tuples(trav(bb)[0])) == sorted(tuple(x) for x in case["contacts_set"])


def test_ray_box_truth_table():
    rb = G["ray_box"]
    box = rb["box"]["lo"] + rb["box"]["up"]
    for c in rb["cases"]:
        assert orc.isintersection(abi.BBOX, abi.F64, box, c["p"], c["d"]) == c["hit"], c


def test_ray_sphere_truth_table():
    rs = G["ray_sphere"]
    for c in rs["unit"]["cases"]:
        assert orc.isintersection(abi.BSPHERE, abi.F64, rs["unit"]["sphere"], c["p"], c["d"]) == c["hit"], c
    for ts in rs["triangle_spheres"]:
        s = orc.volumes_from_triangles(abi.BSPHERE, abi.F64, [ts["tri"]])[0]
        sv = list(s["x"]) + [s["r"]]
        for c in rs["triangle_cases"]:
            d = np.asarray(c["d"])
            assert orc.isintersection(abi.BSPHERE, abi.F64, sv, c["p"], d)
            assert orc.isintersection(abi.BSPHERE, abi.F64, sv, c["p"], -d)


def test_triangle_constructors():
    for c in G["triangle_to_sphere"]["cases"]:
        s = orc.volumes_from_triangles(abi.BSPHERE, abi.F64, [c["tri"]])[0]
        assert np.allclose(s["x"], c["x"]) and np.isclose(s["r"], c["r"])
    for c in G["triangle_to_box"]["cases"]:
        b = orc.volumes_from_triangles(abi.BBOX, abi.F64, [c["tri"]])[0]
        assert np.allclose(b["lo"], c["lo"]) and np.allclose(b["up"], c["up"])


def test_merges():
    ts = abi.make_types(abi.BSPHERE, abi.F64, abi.BSPHERE, abi.F64)
    for c in G["sphere_merge"]["cases"]:
        m = orc.merge(ts, c["a"], c["b"])
        assert np.allclose(m["x"], c["x"], rtol=1e-12) and np.isclose(m["r"], c["r"], rtol=1e-12), c
    tb = abi.make_types(abi.BBOX, abi.F64, abi.BBOX, abi.F64)
    for c in G["box_merge"]["cases"]:
        m = orc.merge(tb, c["a"], c["b"])
        assert np.allclose(m["lo"], c["lo"], rtol=1e-12, atol=0) and np.allclose(m["up"], c["up"], rtol=1e-12), c


@pytest.mark.parametrize("alg", ["lvt", "bfs"])
def test_ray_grid_single_leaf(alg):
    """runtests.jl:1086-1225: single-sphere BVH, grid of origins x 6 axis directions, analytic
    expectation, ORDER-sensitive equality of the hit ray indices."""
    tri = G["ray_grid"]["tri"]
    s = orc.volumes_from_triangles(abi.BSPHERE, abi.F64, [tri])[0]
    x, r = s["x"], float(s["r"])
    types = abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F64)
    bvh = orc.build([list(x) + [r]], types)
    rng = [np.arange(x[k] - r, x[k] + r + 1e-12, 1.0) for k in range(3)]
    pts = np.array([[px, py, pz] for pz in rng[2] for py in rng[1] for px in rng[0]])  # x fastest
    for axis in range(3):
        for sign in (1.0, -1.0):
            d = np.zeros_like(pts)
            d[:, axis] = sign
            trav = orc.traverse_rays_lvt if alg == "lvt" else orc.traverse_rays_bfs
            contacts = trav(bvh, pts, d)[0]
            got = contacts["b"].tolist()
            others = [k for k in range(3) if k != axis]
            exp = []
            for i, p in enumerate(pts):
                behind = p[axis] <= x[axis] if sign > 0 else p[axis] >= x[axis]
                if behind and np.linalg.norm(p[others] - x[others]) <= r:
                    exp.append(i + 1)
                elif (not behind) and np.linalg.norm(p - x) <= r:
                    exp.append(i + 1)
            assert got == exp
            assert set(contacts["a"].tolist()) <= {1}


def test_layouts_match_julia_struct_layout():
    """BSphere{F32}=16, BBox{F32}=24, BoundingVolume{BSphere{F32},Int32,UInt32}=24,
    IndexPair{Int32}=8; BSphere{F64} wrapped = 40 (align 8)."""
    lay = orc.layout_of(abi.make_types())
    assert (lay.volume_bytes, lay.node_bytes, lay.index_off, lay.morton_off, lay.leaf_bytes, lay.pair_bytes) == \
        (16, 24, 16, 20, 24, 8)
    lay = orc.layout_of(abi.make_types(abi.BSPHERE, abi.F64, abi.BBOX, abi.F32))
    assert (lay.volume_bytes, lay.leaf_bytes) == (32, 40)
    for lk in (abi.BSPHERE, abi.BBOX):
        for lf in (abi.F32, abi.F64):
            for it in (abi.I32, abi.I64):
                for mt in (abi.U16, abi.U32, abi.U64):
                    t = abi.make_types(lk, lf, abi.BBOX, abi.F32, it, mt)
                    lay = orc.layout_of(t)
                    dt = abi.leaf_dtype(t)
                    assert dt.itemsize == lay.leaf_bytes
                    assert dt.fields["index"][1] == lay.index_off and dt.fields["morton"][1] == lay.morton_off


def test_merges_into_a_wider_node_type_follow_julias_promotion():
    """build.jl:198-205 builds any node_type, also one WIDER than the leaves' float type; the constructors of merge.jl then
    compute in Julia's promoted type: `BSphere{Float64}(a::BSphere{Float32}, b)` evaluates (b.r - a.r) / length, b.x - a.x
    and length + a.r + b.r in Float32 and everything that meets a T(0.5) / T(1) in Float64 (merge.jl:15-19); the box
    constructors compute min / max / x -+ r in Float32 and convert (merge.jl:30-40, :47-51, :58-81).  The oracle against an
    independent numpy restatement of those rules."""
    rng = np.random.default_rng(5)
    f32, f64 = np.float32, np.float64
    ts = abi.make_types(abi.BSPHERE, abi.F32, abi.BSPHERE, abi.F64)
    tb = abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F64)
    tbb = abi.make_types(abi.BBOX, abi.F32, abi.BBOX, abi.F64)
    for k in range(200):
        a = rng.random(4).astype(f32)
        b = rng.random(4).astype(f32)
        if k % 5 == 0:
            b[3] = f32(5.0)  # a inside b
        if k % 7 == 0:
            a[3] = f32(6.0)  # b inside a
        dx = [f32(a[i] - b[i]) for i in range(3)]
        length = f32(np.sqrt(f32(f32(f32(dx[0] * dx[0]) + f32(dx[1] * dx[1])) + f32(dx[2] * dx[2]))))
        got = orc.merge(ts, a, b)
        if f32(length + a[3]) <= b[3]:
            exp_x, exp_r = [f64(v) for v in b[:3]], f64(b[3])
        elif f32(length + b[3]) <= a[3]:
            exp_x, exp_r = [f64(v) for v in a[:3]], f64(a[3])
        else:
            frac = f64(0.5) * (f64(f32(f32(b[3] - a[3]) / length)) + f64(1))
            exp_x = [f64(a[i]) + frac * f64(f32(b[i] - a[i])) for i in range(3)]
            exp_r = f64(0.5) * f64(f32(f32(length + a[3]) + b[3]))
        assert got["x"].dtype == np.float64
        assert got["x"].tolist() == exp_x and float(got["r"]) == exp_r
        gb = orc.merge(tb, a, b)
        if f32(length + a[3]) <= b[3]:
            lo, up = [f64(f32(b[i] - b[3])) for i in range(3)], [f64(f32(b[i] + b[3])) for i in range(3)]
        elif f32(length + b[3]) <= a[3]:
            lo, up = [f64(f32(a[i] - a[3])) for i in range(3)], [f64(f32(a[i] + a[3])) for i in range(3)]
        else:
            lo = [f64(min(f32(a[i] - a[3]), f32(b[i] - b[3]))) for i in range(3)]
            up = [f64(max(f32(a[i] + a[3]), f32(b[i] + b[3]))) for i in range(3)]
        assert gb["lo"].tolist() == lo and gb["up"].tolist() == up
        ba = np.concatenate([np.minimum(a[:3], b[:3]), np.maximum(a[:3], b[:3]) + f32(0.1)]).astype(f32)
        bb = rng.random(6).astype(f32)
        bb[3:] += bb[:3]
        gbb = orc.merge(tbb, ba, bb)
        assert gbb["lo"].tolist() == [f64(min(ba[i], bb[i])) for i in range(3)]
        assert gbb["up"].tolist() == [f64(max(ba[3 + i], bb[3 + i])) for i in range(3)]
