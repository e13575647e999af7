#!/usr/bin/env python3
"""CPU-side design study for the LVT count pass (no GPU): on the oracle's tree of config 2, what would a wave of 64
consecutive leaves see if it descended the tree with G boxes of 64/G consecutive queries each ("rows") all the way to the
leaf-parent level and then let every row walk its own list of leaf parents?  Prints, per wave (sampled): frontier sizes per
level, leaf parents per row list, steps of the row-synchronous loop, exact (query, parent) candidates.
usage: python tools/sim_lvt_rows.py [n] [waves] [groups]"""
import math
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as orc
from implicitbvh_amd import abi

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
n_waves = int(sys.argv[2]) if len(sys.argv) > 2 else 300
G = int(sys.argv[3]) if len(sys.argv) > 3 else 4
r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
vols = orc.generate_spheres_f32(n, 42, r0=r0)
bvh = orc.build(vols, abi.make_types())
tree = bvh.tree
levels, vl = tree.levels, tree.virtual_leaves
nodes = bvh.nodes
nlo, nup = np.ascontiguousarray(nodes["lo"]), np.ascontiguousarray(nodes["up"])
cx, cr = np.ascontiguousarray(bvh.leaves["volume"]["x"]), np.ascontiguousarray(bvh.leaves["volume"]["r"])
qlo = cx - cr[:, None]
qup = cx + cr[:, None]


def popcount(v):
    return bin(v).count("1")


def num_real(level):
    return (1 << (level - 1)) - (vl >> (levels - level))


def first_mem(level):
    v = vl >> (levels - (level - 1))
    return (1 << (level - 1)) - (2 * v - popcount(v)) - 1


lp = levels - 1
rng = np.random.default_rng(1)
waves = rng.choice(n // 64, size=min(n_waves, n // 64), replace=False)
rows = 64 // G
front = {l: [] for l in range(7, lp + 1)}
sum_lists, max_lists, cands, steps_bal, parents_any = [], [], [], [], []
for w in waves:
    i0 = int(w) * 64
    blo = np.stack([qlo[i0 + g * rows: i0 + (g + 1) * rows].min(0) for g in range(G)])
    bup = np.stack([qup[i0 + g * rows: i0 + (g + 1) * rows].max(0) for g in range(G)])
    idx = np.arange(num_real(7), dtype=np.int64) + (1 << 6)  # implicit indices at level 7
    for lvl in range(7, lp + 1):
        mem = idx - (1 << (lvl - 1)) + first_mem(lvl)
        lo, up = nlo[mem], nup[mem]
        # node (level lvl, implicit idx): its last leaf position (0-based) = ((idx+1) << (levels-lvl)) - leaf_first - 1
        last_leaf = ((idx + 1) << (levels - lvl)) - (1 << (levels - 1)) - 1
        touch = np.zeros((G, len(idx)), bool)
        for g in range(G):
            t = np.all(blo[g] <= up, 1) & np.all(bup[g] >= lo, 1)
            t &= last_leaf > i0 + g * rows  # self prune: something to the right of the row's first item
            touch[g] = t
        hit = touch.any(0)
        front[lvl].append(len(idx))
        if lvl == lp:
            sizes = touch.sum(1)
            sum_lists.append(int(sizes.sum()))
            max_lists.append(int(sizes.max()))
            parents_any.append(int(hit.sum()))
            # exact candidates: query box vs parent box, 2p+1 > item
            c = 0
            pidx = idx - (1 << (lp - 1))
            for g in range(G):
                sel = touch[g]
                if not sel.any():
                    continue
                for qi in range(i0 + g * rows, i0 + (g + 1) * rows):
                    t = np.all(qlo[qi] <= up[sel], 1) & np.all(qup[qi] >= lo[sel], 1) & (2 * pidx[sel] + 1 > qi)
                    c += int(t.sum())
            cands.append(c)
        else:
            kids = np.stack([2 * idx[hit], 2 * idx[hit] + 1], 1).reshape(-1)
            kids = kids[kids - (1 << lvl) < num_real(lvl + 1)]
            idx = kids
print(f"n = {n}, levels = {levels}, groups of {rows} queries: {G}, waves sampled: {len(waves)}")
tot_chunks = 0.0
for lvl in range(7, lp + 1):
    a = np.array(front[lvl])
    ch = np.ceil(a / 64).mean()
    tot_chunks += ch
    print(f"  level {lvl:2d}: frontier mean {a.mean():7.1f}  p95 {np.percentile(a, 95):7.1f}  max {a.max():5d}  64-lane steps {ch:5.2f}")
print(f"  descent 64-lane steps per wave: {tot_chunks:.1f}")
for nm, a in (("leaf parents touching any row", parents_any), ("sum of row lists", sum_lists), ("longest row list (= loop steps)", max_lists),
              ("exact (query, parent) candidates", cands)):
    a = np.array(a)
    print(f"  {nm:36s} mean {a.mean():7.1f}  p95 {np.percentile(a, 95):7.1f}  max {a.max():5d}")
