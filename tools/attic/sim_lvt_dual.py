#!/usr/bin/env python3
"""CPU-side design study for the LVT count pass (no GPU): a wave-local DUAL descent.  A wave = 64 consecutive leaves with a
binary hierarchy over its lanes (64 -> 32 -> ... -> 1 queries); pairs (query group Q, tree node T) that touch are expanded
level-synchronously: first only T is split (Q = all 64 queries), the last six levels split both sides, ending in (single
query, leaf parent) pairs = the candidates of the leaf-test step.  Children are tested when they are generated; queues hold
passing pairs only.  Prints per level: pairs popped, tests, 64-lane steps; and the distribution over waves.
usage: python tools/sim_lvt_dual.py [n] [waves|all] [mode]   mode: fixed | split (adaptive first split of the 64 lanes)"""
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
n_waves = sys.argv[2] if len(sys.argv) > 2 else "300"
mode = sys.argv[3] if len(sys.argv) > 3 else "fixed"
early = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # start the dual levels this many T levels earlier (Q then waits at single queries)
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
leaf_first = 1 << (levels - 1)
all_waves = np.arange((n + 63) // 64)
if n_waves != "all":
    rng = np.random.default_rng(1)
    all_waves = rng.choice(all_waves, size=min(int(n_waves), len(all_waves)), replace=False)

T_ONLY_LAST = lp - 6 - early  # T level whose nodes are paired with Q depth 0 before the dual levels start
per_level_tests = {}
per_level_pairs = {}
tot_tests, tot_steps, cand = [], [], []
for w in all_waves:
    i0 = int(w) * 64
    cnt = min(64, n - i0)
    lo = np.full((64, 3), np.inf, np.float32)
    up = np.full((64, 3), -np.inf, np.float32)
    lo[:cnt], up[:cnt] = qlo[i0:i0 + cnt], qup[i0:i0 + cnt]
    # Q hierarchy.  fixed: Qlo[d] has 2^d boxes (depth 0 = all lanes).  split: the wave is cut at the lane k that minimises
    # the half-area sum of box[0..k] + box[k+1..63]; node (d, i, side) = lanes of the aligned group i of size 64 >> (d-1) on
    # that side of the cut, stored at index 2 i + side of depth d (d = 1 .. 7); empty nodes have inverted boxes.
    if mode == "fixed":
        Qlo, Qup = [None] * 7, [None] * 7
        Qlo[6], Qup[6] = lo, up
        for d in range(5, -1, -1):
            Qlo[d] = np.minimum(Qlo[d + 1][0::2], Qlo[d + 1][1::2])
            Qup[d] = np.maximum(Qup[d + 1][0::2], Qup[d + 1][1::2])
        first_item = {d: i0 + np.arange(1 << d) * (64 >> d) for d in range(7)}  # first item of each Q node
        QD = 6
    else:
        def harea(l, u):
            d = np.maximum(u - l, 0)
            return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 0] * d[..., 2]
        plo, pup = np.minimum.accumulate(lo, 0), np.maximum.accumulate(up, 0)
        slo, sup = np.minimum.accumulate(lo[::-1], 0)[::-1], np.maximum.accumulate(up[::-1], 0)[::-1]
        cost = harea(plo, pup).astype(np.float64)
        cost[:-1] += harea(slo[1:], sup[1:])
        k = int(np.argmin(cost))
        side = (np.arange(64) > k).astype(int)
        Qlo, Qup, first_item = [None] * 8, [None] * 8, {}
        for d in range(1, 8):
            g = 64 >> (d - 1)
            ng = 64 // g
            L = np.full((2 * ng, 3), np.inf, np.float32)
            U = np.full((2 * ng, 3), -np.inf, np.float32)
            F = np.full(2 * ng, 1 << 40, np.int64)
            for i in range(ng):
                for sd in (0, 1):
                    sel = np.arange(i * g, (i + 1) * g)
                    sel = sel[side[sel] == sd]
                    if len(sel):
                        L[2 * i + sd], U[2 * i + sd] = lo[sel].min(0), up[sel].max(0)
                        F[2 * i + sd] = i0 + sel[0]
            Qlo[d], Qup[d], first_item[d] = L, U, F
        QD = 7
    # start: pairs (Q0, T) for every T of level 7 that passes
    tl = 7
    t = np.arange(num_real(tl), dtype=np.int64) + (1 << (tl - 1))
    if mode == "fixed":
        qd = 0
        q = np.zeros(len(t), np.int64)
    else:
        qd = 1
        t = np.repeat(t, 2)
        q = np.tile(np.array([0, 1]), len(t) // 2)
    wt, ws = 0, 0

    def test(qd, q, tl, t):
        mem = t - (1 << (tl - 1)) + first_mem(tl)
        ok = np.all(Qlo[qd][q] <= nup[mem], 1) & np.all(Qup[qd][q] >= nlo[mem], 1)
        last_leaf = ((t + 1) << (levels - tl)) - leaf_first - 1
        return ok & (last_leaf > first_item[qd][q])

    keep = test(qd, q, tl, t)
    wt += len(t)
    ws += 1
    q, t = q[keep], t[keep]
    while tl < lp or qd < QD:
        split_q = qd < QD and tl >= T_ONLY_LAST
        split_t = tl < lp
        key = (tl, qd)
        per_level_pairs.setdefault(key, []).append(len(t))
        ws += max(1, math.ceil(len(t) / 64)) if len(t) else 0
        # children, T-major order
        ts = [2 * t, 2 * t + 1] if split_t else [t]
        if mode == "fixed":
            qs = [2 * q, 2 * q + 1] if split_q else [q]
        else:  # (i, side) -> (2i, side), (2i+1, side)
            qs = [4 * (q >> 1) + (q & 1), 4 * (q >> 1) + 2 + (q & 1)] if split_q else [q]
        nt, nq = [], []
        for tc in ts:
            for qc in qs:
                nt.append(tc)
                nq.append(qc)
        nt, nq = np.stack(nt, 1).reshape(-1), np.stack(nq, 1).reshape(-1)
        tl2, qd2 = tl + (1 if split_t else 0), qd + (1 if split_q else 0)
        real = (nt - (1 << (tl2 - 1))) < num_real(tl2)
        nt, nq = nt[real], nq[real]
        keep = test(qd2, nq, tl2, nt)
        per_level_tests.setdefault(key, []).append(len(nt))
        wt += len(nt)
        q, t, tl, qd = nq[keep], nt[keep], tl2, qd2
    tot_tests.append(wt)
    tot_steps.append(ws)
    cand.append(len(t))
print(f"n = {n}, levels = {levels}, waves: {len(all_waves)}, mode {mode}, dual levels start at T level {T_ONLY_LAST}")
for key in sorted(per_level_pairs):
    a, b = np.array(per_level_pairs[key]), np.array(per_level_tests[key])
    print(f"  pop at (T level {key[0]:2d}, Q depth {key[1]}): pairs mean {a.mean():7.1f} p99 {np.percentile(a, 99):7.1f} max {a.max():6d} | child tests mean {b.mean():7.1f} max {b.max():6d}")
for nm, a in (("lane-level tests per wave", tot_tests), ("64-lane steps per wave", tot_steps), ("candidates", cand)):
    a = np.array(a)
    print(f"  {nm:28s} mean {a.mean():8.1f}  p95 {np.percentile(a, 95):8.1f}  p99 {np.percentile(a, 99):8.1f}  max {a.max():7d}")
print(f"  tests per leaf: {np.mean(tot_tests) / 64:.1f}")
if n_waves == "all":
    # makespan model: waves dealt in index order to `slots` wave slots, a wave's time proportional to its steps
    import heapq
    steps = np.array(tot_steps, float)
    for slots in (7168, 6144):
        h = [0.0] * slots
        heapq.heapify(h)
        for s in steps:
            heapq.heappush(h, heapq.heappop(h) + s)
        print(f"  makespan model, {slots} wave slots: ideal {steps.sum() / slots:.1f} steps, greedy {max(h):.1f} steps, worst wave {steps.max():.0f}")
    a = np.array(tot_tests)
    print("  waves with > 2x / 4x / 8x the mean tests:", int((a > 2 * a.mean()).sum()), int((a > 4 * a.mean()).sum()), int((a > 8 * a.mean()).sum()))
