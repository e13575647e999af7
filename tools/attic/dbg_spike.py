"""Run ON THE GPU BOX: the latency of the ONE rebuild on which a `cache=` chain's input changes from a uniform cloud to a
clustered one (the skew hint of the chain still says "uniform").  usage: python tools/dbg_spike.py [n] [spare]
spare = 0: the changed input meets sort_levels = 0 (every crowded cell takes the one-workgroup slow path);
spare = 1: at least one extra level on every cached build (api.SPARE_OCCUPANCY = 0);
spare = 2: what the Python mirror does by default — a spare level only when the previous build's fullest cell was beyond
           api.SPARE_OCCUPANCY / 128 of a finish workgroup's capacity (an abrupt change from a comfortable uniform cloud
           then meets sort_levels = 0 once); also prints a GRADUAL change: the cloud contracts by 2 % per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import api

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
spare = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if spare == 1:
    api.SPARE_OCCUPANCY = 0
g = torch.Generator(device="cuda").manual_seed(3)
uniform = torch.rand((n, 4), generator=g, device="cuda") * torch.tensor([1, 1, 1, 1e-4], device="cuda")
one = torch.empty((n, 4), device="cuda")
one[:, :3] = 0.5 + 0.001 * torch.randn((n, 3), generator=g, device="cuda")
one[:, 3] = 1e-4
one[0, :3] = 100.0  # a far outlier: everything else shares a handful of coarse cells
c8 = torch.rand((8, 3), generator=g, device="cuda")
clusters = torch.empty((n, 4), device="cuda")
clusters[:, :3] = c8[torch.randint(0, 8, (n,), generator=g, device="cuda")] + 0.004 * torch.randn((n, 3), generator=g, device="cuda")
clusters[:, 3] = 1e-4


def timed(v, cache):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = ibvh.BVH(v, cache=cache)
    torch.cuda.synchronize()
    return b, (time.perf_counter() - t0) * 1e3


for name, changed in (("one cluster + outlier", one), ("8 gaussian clusters", clusters)):
    b = None
    for _ in range(4):
        b, t_u = timed(uniform, b)
    if not spare:
        b._skew[0] = 0
        saved = api.abi.MAX_SORT_LEVELS
        # (emulate the round-2 mirror: hint 0 -> no extra level)
        orig = api.BVH.__init__
    times = []
    for k in range(4):
        if not spare and k == 0:
            # force sort_levels = 0 for this one build: a cold build with COLD_SORT_LEVELS = 0 on the cached buffers
            keep = api.COLD_SORT_LEVELS
            api.COLD_SORT_LEVELS = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b2 = ibvh.BVH(changed)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
            api.COLD_SORT_LEVELS = keep
            b = b2
            continue
        b, t = timed(changed, b)
        times.append(t)
    m = b.leaves.morton
    assert bool((m[1:] >= m[:-1]).all())
    print(f"n={n} spare={spare} {name}: uniform step {t_u:.2f} ms; steps after the change: " + " ".join(f"{t:.2f}" for t in times) + f" ms; hint {b._skew[0]}")

if spare == 2:
    # gradual: every leaf moves 2 % of the way to the cloud's centre per step (density x 1.06 per step)
    b = None
    v = uniform.clone()
    worst = 0.0
    log = []
    for step in range(60):
        v[:, :3] = 0.5 + (v[:, :3] - 0.5) * 0.98
        v[0, :3] = 0.0   # two fixed outliers keep the Morton grid's extent: the cloud really gets denser in it
        v[1, :3] = 1.0
        b, t = timed(v, b)
        if step >= 2:
            worst = max(worst, t)
        log.append((t, b._skew[0], b._skew.occupancy()))
    m = b.leaves.morton
    assert bool((m[1:] >= m[:-1]).all())
    print(f"n={n} gradual contraction, 60 steps: worst step {worst:.2f} ms; (ms, levels, occupancy/128) every 6th step: " +
          " ".join(f"({t:.2f},{l},{o})" for t, l, o in log[::6]))
