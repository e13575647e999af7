#!/usr/bin/env python3
"""CPU-side design study (no GPU): how many cut-level (128-leaf) subtrees and leaf parents a wave of 64 consecutive sorted
leaves touches with ONE / TWO / THREE / FOUR query boxes (contiguous lane ranges; two = the kernel's split at the lane that
minimises the half-area sum; three / four = the costlier part split again the same way).
usage: python tools/sim_wave_boxes.py [n] [waves sampled]"""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as orc
from implicitbvh_amd import abi

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 400
r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
bvh = orc.build(orc.generate_spheres_f32(n, 42, r0=r0), abi.make_types())
levels, vl = bvh.tree.levels, bvh.tree.virtual_leaves
nlo, nup = np.ascontiguousarray(bvh.nodes["lo"]), np.ascontiguousarray(bvh.nodes["up"])
cx, cr = np.ascontiguousarray(bvh.leaves["volume"]["x"]), np.ascontiguousarray(bvh.leaves["volume"]["r"])
qlo, qup = cx - cr[:, None], cx + cr[:, None]
pc = lambda v: bin(v).count("1")
num_real = lambda l: (1 << (l - 1)) - (vl >> (levels - l))
def first_mem(l):
    v = vl >> (levels - (l - 1))
    return (1 << (l - 1)) - (2 * v - pc(v)) - 1
cut, lp = levels - 7, levels - 1
clo, cup = nlo[first_mem(cut):first_mem(cut) + num_real(cut)], nup[first_mem(cut):first_mem(cut) + num_real(cut)]
plo, pup = nlo[first_mem(lp):first_mem(lp) + num_real(lp)], nup[first_mem(lp):first_mem(lp) + num_real(lp)]
def harea(l, u):
    d = np.maximum(u - l, 0)
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 0] * d[..., 2]
def best_split(lo, up):
    """lane k minimising area(0..k) + area(k+1..); returns (k, cost)"""
    if len(lo) < 2:
        return None, np.inf
    pl, pu = np.minimum.accumulate(lo, 0), np.maximum.accumulate(up, 0)
    sl, su = np.minimum.accumulate(lo[::-1], 0)[::-1], np.maximum.accumulate(up[::-1], 0)[::-1]
    cost = harea(pl[:-1], pu[:-1]) + harea(sl[1:], su[1:])
    k = int(np.argmin(cost))
    return k, float(cost[k])
def parts(lo, up, m):
    segs = [(0, len(lo))]
    while len(segs) < m:
        gains = []
        for (a, b) in segs:
            k, c = best_split(lo[a:b], up[a:b])
            whole = float(harea(lo[a:b].min(0), up[a:b].max(0)))
            gains.append((whole - c if k is not None else -1, a, b, k))
        g, a, b, k = max(gains)
        if k is None or g <= 0:
            break
        segs.remove((a, b))
        segs += [(a, a + k + 1), (a + k + 1, b)]
    return [(lo[a:b].min(0), up[a:b].max(0)) for a, b in sorted(segs)]
rng = np.random.default_rng(1)
waves = rng.choice((n + 63) // 64, size=nw, replace=False)
res = {m: {"sub": [], "par": []} for m in (1, 2, 3, 4)}
for w in waves:
    i0 = int(w) * 64
    lo, up = qlo[i0:i0 + 64], qup[i0:i0 + 64]
    for m in res:
        boxes = parts(lo, up, m)
        t = np.zeros(len(clo), bool)
        for bl, bu in boxes:
            t |= np.all((clo <= bu) & (cup >= bl), axis=1)
        t &= (np.arange(len(clo)) + 1) * 128 > i0            # self prune at the cut level
        res[m]["sub"].append(int(t.sum()))
        npar = 0
        for c in np.nonzero(t)[0]:
            a, b = c * 64, min((c + 1) * 64, len(plo))
            tp = np.zeros(b - a, bool)
            for bl, bu in boxes:
                tp |= np.all((plo[a:b] <= bu) & (pup[a:b] >= bl), axis=1)
            tp &= 2 * (np.arange(a, b)) + 1 > i0
            npar += int(tp.sum())
        res[m]["par"].append(npar)
for m in res:
    s, p = np.array(res[m]["sub"]), np.array(res[m]["par"])
    print(f"{m} box(es): subtrees per wave mean {s.mean():.1f} p95 {np.percentile(s,95):.0f} max {s.max()}; leaf parents touching the boxes mean {p.mean():.1f} p95 {np.percentile(p,95):.0f}")
