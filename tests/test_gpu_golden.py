"""The reference's own known answers pushed STRAIGHT through the HIP library (C ABI via the Python mirror) — no oracle
in between: ray-box (9 cases) and ray-sphere (16 cases) truth tables (test/runtests.jl:322-468), the single-leaf ray grid
(:1086-1225), triangle -> bounding-volume expectations (:186-210, :263-276) and merge expectations (:222-252, :288-318).
The fixtures are tests/golden/reference_known_answers.json (data only, each entry cites its source lines).

How a single predicate is reached through entry points that only traverse trees: a BVH of ONE leaf has no nodes, so
`traverse_rays` reports leaf 1 for ray i exactly when isintersection(leaf, p_i, d_i) holds; a BVH of TWO leaves has one
node — the merge of the two leaves — which `bvh.nodes[0]` exposes; a one-leaf BBox BVH built from a box is the ray-box
predicate itself (leaf test), and the same box as the ROOT NODE of a two-leaf tree is the node-level slab test."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
DT = {"f32": (np.float32, torch.float32), "f64": (np.float64, torch.float64)}


def cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ray_hits(bvh, pts, dirs, alg, npdt):
    """-> set of 1-based ray indices that hit anything, and the raw contact rows"""
    p = cuda(np.asarray(pts, npdt)).t()
    d = cuda(np.asarray(dirs, npdt)).t()
    t = ibvh.traverse_rays(bvh, p, d, alg)
    c = t.contacts.cpu().numpy()
    return c


@pytest.mark.parametrize("flt", ["f32", "f64"])
@pytest.mark.parametrize("alg", ["lvt", "bfs"])
def test_ray_box_truth_table_through_the_library(flt, alg):
    """runtests.jl:322-379: isintersection(BBox, p, d) for the nine (p, d) cases, as the LEAF test of a one-leaf BVH."""
    npdt, tdt = DT[flt]
    box = G["ray_box"]["box"]
    vol = np.array([box["lo"] + box["up"]], npdt)
    bvh = ibvh.BVH(cuda(vol), ibvh.BBox(tdt))
    cases = G["ray_box"]["cases"]
    a = ibvh.LVTTraversal() if alg == "lvt" else ibvh.BFSTraversal()
    c = ray_hits(bvh, [k["p"] for k in cases], [k["d"] for k in cases], a, npdt)
    got = sorted(c[:, 1].tolist())
    assert got == [i + 1 for i, k in enumerate(cases) if k["hit"]]
    assert set(c[:, 0].tolist()) <= {1}


@pytest.mark.parametrize("alg", ["lvt", "bfs"])
def test_ray_box_truth_table_at_node_level(alg):
    """The same nine cases against the box as the ROOT NODE: two leaves whose merge is exactly the unit box (two half
    boxes), every hit of the root must then reach a leaf test, every miss must report nothing."""
    box = G["ray_box"]["box"]
    lo, up = np.array(box["lo"], np.float32), np.array(box["up"], np.float32)
    mid = (lo[0] + up[0]) / 2
    vols = np.array([[lo[0], lo[1], lo[2], mid, up[1], up[2]], [mid, lo[1], lo[2], up[0], up[1], up[2]]], np.float32)
    bvh = ibvh.BVH(cuda(vols), ibvh.BBox(torch.float32))
    root = bvh.nodes.cpu().numpy().reshape(-1)[:6]
    assert root.tolist() == lo.tolist() + up.tolist()  # box + box merge (merge.jl:30-43) = the unit box
    cases = G["ray_box"]["cases"]
    a = ibvh.LVTTraversal() if alg == "lvt" else ibvh.BFSTraversal()
    c = ray_hits(bvh, [k["p"] for k in cases], [k["d"] for k in cases], a, np.float32)
    hit_rays = sorted(set(c[:, 1].tolist()))
    assert hit_rays == [i + 1 for i, k in enumerate(cases) if k["hit"]]  # a ray that hits the union hits one of the halves


@pytest.mark.parametrize("flt", ["f32", "f64"])
@pytest.mark.parametrize("alg", ["lvt", "bfs"])
def test_ray_sphere_truth_table_through_the_library(flt, alg):
    """runtests.jl:382-468: eight cases on the unit sphere, then two triangle spheres x two origins x (d, -d): all hit."""
    npdt, tdt = DT[flt]
    a = ibvh.LVTTraversal() if alg == "lvt" else ibvh.BFSTraversal()
    unit = G["ray_sphere"]["unit"]
    bvh = ibvh.BVH(cuda(np.array([unit["sphere"]], npdt)), ibvh.BBox(tdt))
    cases = unit["cases"]
    c = ray_hits(bvh, [k["p"] for k in cases], [k["d"] for k in cases], a, npdt)
    assert sorted(c[:, 1].tolist()) == [i + 1 for i, k in enumerate(cases) if k["hit"]]
    # the spheres of two triangles (made by the library's triangle kernel), both origins, d and -d: eight hits each
    for ts in G["ray_sphere"]["triangle_spheres"]:
        vol = ibvh.bounding_volumes_from_triangles(cuda(np.array([ts["tri"]], npdt)))
        b = ibvh.BVH(vol, ibvh.BBox(tdt))
        pts, dirs = [], []
        for k in G["ray_sphere"]["triangle_cases"]:
            for sgn in (1.0, -1.0):
                pts.append(k["p"])
                dirs.append([sgn * x for x in k["d"]])
        c = ray_hits(b, pts, dirs, a, npdt)
        assert sorted(c[:, 1].tolist()) == list(range(1, len(pts) + 1))


@pytest.mark.parametrize("alg", ["lvt", "bfs"])
def test_ray_grid_single_leaf_through_the_library(alg):
    """runtests.jl:1086-1225: single-sphere BVH (BBox{Float64} nodes), origins on the grid (x-r):1:(x+r) per axis (x fastest),
    six axis directions; the hit list is analytic and, for the leaf-vs-tree walk, ORDER-sensitive."""
    tri = np.array([G["ray_grid"]["tri"]], np.float64)
    vol = ibvh.bounding_volumes_from_triangles(cuda(tri))
    s = vol.cpu().numpy()[0]
    x, r = s[:3], float(s[3])
    bvh = ibvh.BVH(vol, ibvh.BBox(torch.float64))
    rng = [np.arange(x[k] - r, x[k] + r + 1e-12, 1.0) for k in range(3)]
    pts = np.array([[px, py, pz] for pz in rng[2] for py in rng[1] for px in rng[0]])
    a = ibvh.LVTTraversal() if alg == "lvt" else ibvh.BFSTraversal()
    for axis in range(3):
        for sign in (1.0, -1.0):
            d = np.zeros_like(pts)
            d[:, axis] = sign
            c = ray_hits(bvh, pts, d, a, np.float64)
            others = [k for k in range(3) if k != axis]
            exp = []
            for i, p in enumerate(pts):
                behind = p[axis] <= x[axis] if sign > 0 else p[axis] >= x[axis]
                if behind and np.linalg.norm(p[others] - x[others]) <= r:
                    exp.append(i + 1)
                elif (not behind) and np.linalg.norm(p - x) <= r:
                    exp.append(i + 1)
            got = c[:, 1].tolist()
            assert (got if alg == "lvt" else sorted(got)) == exp
            assert set(c[:, 0].tolist()) <= {1}


@pytest.mark.parametrize("flt", ["f32", "f64"])
def test_triangle_constructors_through_the_library(flt):
    """runtests.jl:186-210 (BSphere(p1, p2, p3)) and :263-276 (BBox(p1, p2, p3)) through ibvh_volumes_from_triangles."""
    npdt, tdt = DT[flt]
    tol = 1e-6 if flt == "f32" else 1e-12
    for k in G["triangle_to_sphere"]["cases"]:
        v = ibvh.bounding_volumes_from_triangles(cuda(np.array([k["tri"]], npdt))).cpu().numpy()[0]
        assert np.allclose(v[:3], k["x"], atol=tol) and abs(float(v[3]) - k["r"]) <= tol
    for k in G["triangle_to_box"]["cases"]:
        v = ibvh.bounding_volumes_from_triangles(cuda(np.array([k["tri"]], npdt)), ibvh.BBox(tdt)).cpu().numpy()[0]
        assert v[:3].tolist() == k["lo"] and v[3:].tolist() == k["up"]  # min / max of the corners: exact


@pytest.mark.parametrize("flt", ["f32", "f64"])
def test_merges_through_the_library(flt):
    """runtests.jl:222-252 (sphere + sphere) and :288-318 (box + box): the single node of a two-leaf BVH is the merge.  The
    build sorts its leaves by Morton code first; both merges are symmetric in their arguments up to the containment
    branches, which the golden cases cover in both orders."""
    npdt, tdt = DT[flt]
    tol = 1e-6 if flt == "f32" else 1e-12
    for k in G["sphere_merge"]["cases"]:
        b = ibvh.BVH(cuda(np.array([k["a"], k["b"]], npdt)), ibvh.BSphere(tdt))
        node = b.nodes.cpu().numpy().reshape(-1)[:4]
        assert np.allclose(node[:3], k["x"], atol=tol) and abs(float(node[3]) - k["r"]) <= tol
    for k in G["box_merge"]["cases"]:
        b = ibvh.BVH(cuda(np.array([k["a"], k["b"]], npdt)), ibvh.BBox(tdt))
        node = b.nodes.cpu().numpy().reshape(-1)[:6]
        assert node[:3].tolist() == [npdt(v) for v in k["lo"]] and node[3:].tolist() == [npdt(v) for v in k["up"]]
