#!/usr/bin/env python3
"""CPU-side design study (no GPU) for the shared descent of the LVT count pass: for blocks of 2^s consecutive sorted leaves
(= the tree's own node s levels above the leaves), how many cut-level (128-leaf) subtrees touch the block's boxes — the size
of the per-block list the waves of the block would filter instead of descending on their own — for 1 / 2 / 4 boxes per block
(the node itself, its children, its grandchildren); next to it the per-wave list (two boxes per wave, as the kernel splits).
usage: python tools/sim_block_frontier.py [n] [blocks sampled]"""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as orc
from implicitbvh_amd import abi

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 200
r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
bvh = orc.build(orc.generate_spheres_f32(n, 42, r0=r0), abi.make_types())
levels, vl = bvh.tree.levels, bvh.tree.virtual_leaves
nlo, nup = np.ascontiguousarray(bvh.nodes["lo"]), np.ascontiguousarray(bvh.nodes["up"])
pc = lambda v: bin(v).count("1")
num_real = lambda l: (1 << (l - 1)) - (vl >> (levels - l))
def first_mem(l):
    v = vl >> (levels - (l - 1))
    return (1 << (l - 1)) - (2 * v - pc(v)) - 1
cut = levels - 7
c0, cn = first_mem(cut), num_real(cut)
clo, cup = nlo[c0:c0 + cn], nup[c0:c0 + cn]
def touching(lo, up):  # cut nodes touching box (lo, up)
    return np.all((clo <= up) & (cup >= lo), axis=1)
rng = np.random.default_rng(1)
for s in (9, 10, 11):
    lvl = levels - s  # block node level
    nblk = num_real(lvl)
    pick = rng.choice(nblk, size=min(nb, nblk), replace=False)
    res = {1: [], 2: [], 4: []}
    for b in pick:
        right_of = (np.arange(cn) + 1) * 128 > b * (1 << s)   # self prune: subtree ends right of the block's first leaf
        for k, d in ((1, 0), (2, 1), (4, 2)):
            l2 = lvl + d
            m = np.zeros(cn, bool)
            for j in range(k):
                i = b * k + j
                if i < num_real(l2):
                    m |= touching(nlo[first_mem(l2) + i], nup[first_mem(l2) + i])
            res[k].append(int((m & right_of).sum()))
    print(f"block 2^{s} leaves:", {k: (round(float(np.mean(v)), 1), int(np.percentile(v, 95)), int(np.max(v))) for k, v in res.items()}, "(mean, p95, max) by boxes per block")
