"""Per-level visit counts of the ray walk on config 3's tree (CPU, oracle build + numpy expansion): where the steps of
`traverse_rays` are spent, and how many (ray, subtree) items a cut at level K would produce.  Development tool."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle_lib as orc  # noqa: E402
from implicitbvh_amd import abi  # noqa: E402
from implicitbvh_amd.synthetic import random_rays, torus_mesh  # noqa: E402


def main():
    nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    tris = torus_mesh()
    types = abi.make_types(abi.BSPHERE, abi.F32, abi.BBOX, abi.F32)
    t0 = time.time()
    vols = orc.volumes_from_triangles(abi.BSPHERE, abi.F32, tris)
    v = np.ascontiguousarray(vols).view(np.float32).reshape(len(vols), -1)
    o = orc.build(v, types)
    print("build", round(time.time() - t0, 1), "s; levels", o.tree.levels, "real leaves", o.tree.real_leaves)
    lo, hi = v[:, :3].min(0), v[:, :3].max(0)
    p, d = random_rays(1_000_000, lo, hi, seed=43)
    p, d = p[:nrays].astype(np.float64), d[:nrays].astype(np.float64)
    inv = 1.0 / d
    nodes = np.ascontiguousarray(o.nodes).view(np.float32).reshape(len(o.nodes), -1)[:, :6].astype(np.float64)
    lv = np.ascontiguousarray(o.leaves["volume"]).view(np.float32).reshape(len(o.leaves), -1)[:, :4].astype(np.float64)
    skips = orc.compute_skips(o.tree)
    L = o.tree.levels
    ray = np.arange(nrays)
    node = np.ones(nrays, np.int64)  # implicit index, level 1
    per_level, per_ray_steps = [], np.zeros(nrays, np.int64)
    items_at = {}
    CUT = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    item = None
    for lvl in range(1, L):
        real_at = [orc.level_indices(o.tree, lvl)]
        first = 1 << (lvl - 1)
        nreal = (o.tree.real_leaves + (1 << (L - lvl)) - 1) >> (L - lvl)
        ok = (node - first) < nreal
        ray, node = ray[ok], node[ok]
        mem = node - 1 - skips[lvl - 1]
        b = nodes[mem]
        t1 = (b[:, :3] - p[ray]) * inv[ray]
        t2 = (b[:, 3:] - p[ray]) * inv[ray]
        tmin = np.minimum(t1, t2).max(1)
        tmax = np.maximum(t1, t2).min(1)
        hit = (tmax >= np.maximum(tmin, 0.0))
        per_level.append((lvl, len(ray), int(hit.sum())))
        items_at[lvl] = int(hit.sum())
        np.add.at(per_ray_steps, ray[hit], 1)
        ray, node = ray[hit], node[hit]
        if lvl > CUT:
            item = item[ok][hit]
            np.add.at(item_steps, item, 1)
        if lvl == CUT:
            j = node - first
            bc = np.bincount(j, minlength=nreal)
            print(f"cut {CUT}: items/ray {len(j) / nrays:.2f}; items per subtree (scaled to 1e6 rays): mean {bc.mean() * 1e6 / nrays:.0f} "
                  f"p50 {np.percentile(bc, 50) * 1e6 / nrays:.0f} p99 {np.percentile(bc, 99) * 1e6 / nrays:.0f} max {bc.max() * 1e6 / nrays:.0f}")
            item = np.arange(len(j))
            item_steps = np.zeros(len(j), np.int64)
            item_sub = j.copy()
        if lvl >= CUT:
            item = np.repeat(item, 2)
        ray = np.repeat(ray, 2)
        node = np.repeat(node * 2, 2)
        node[1::2] += 1
    first = 1 << (L - 1)
    ok = (node - first) < o.tree.real_leaves
    ray, node = ray[ok], node[ok]
    s = lv[node - first]
    oc = p[ray] - s[:, :3]
    dd = d[ray]
    a = (dd * dd).sum(1)
    bq = 2 * (oc * dd).sum(1)
    c = (oc * oc).sum(1) - s[:, 3] ** 2
    disc = bq * bq - 4 * a * c
    hit = (disc >= 0) & ((-bq + np.sqrt(np.maximum(disc, 0))) >= 0)
    print("leaf tests", len(ray) / nrays, "hits/ray", hit.sum() / nrays)
    tot = 0
    for lvl, tested, h in per_level:
        tot += h
        print(f"level {lvl:2d}: tests/ray {tested / nrays:8.2f}  visited/ray {h / nrays:8.2f}  cumulative steps/ray {tot / nrays:8.2f}")
    print("steps per item (below the cut, node levels only): mean %.1f p50 %d p90 %d p99 %d max %d" % ((item_steps.mean(),) + tuple(np.percentile(item_steps, [50, 90, 99, 100]))))
    sub_steps = np.bincount(item_sub, weights=item_steps)
    print("node steps per subtree (scaled to 1e6 rays): mean %.0f p99 %.0f max %.0f" % (sub_steps.mean() * 1e6 / nrays, np.percentile(sub_steps, 99) * 1e6 / nrays, sub_steps.max() * 1e6 / nrays))
    q = np.percentile(per_ray_steps, [50, 90, 99, 99.9, 100])
    print("steps per ray percentiles 50/90/99/99.9/max", q)


if __name__ == "__main__":
    main()
