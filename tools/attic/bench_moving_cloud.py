"""Run ON THE GPU BOX: the reference's time-stepping loop (build.jl:109-126, README.md:84-95) on a MOVING cloud — row f3 of
SURVEY.md §8: every step perturbs each centre by at most `--cells` cells of the 1024^3 Morton grid, feeds the previous
step's Morton-ordered leaves back in (so the input is NEARLY sorted, not sorted) and rebuilds through `cache=`.
Prints the build time per step next to the same cloud handed over in random order, how disordered the input really is
(leaves whose key is smaller than their predecessor's; leaves that change their coarse cell) and the per-kernel times.
usage: python tools/bench_moving_cloud.py [n] [--cells 1.0] [--steps 20]"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from bench import collect_profile

ap = argparse.ArgumentParser()
ap.add_argument("n", nargs="?", type=float, default=1e6)
ap.add_argument("--cells", type=float, default=1.0)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
n = int(args.n)
r0 = 0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3)
vols = ibvh.generate_spheres(n, 42, r0=r0)
g = torch.Generator(device="cuda").manual_seed(11)
step = args.cells / 1024.0  # one cell of the 10-bit-per-axis grid of UInt32 codes on the unit cube


def timed_build(v, cache):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = ibvh.BVH(v, cache=cache)
    torch.cuda.synchronize()
    return b, (time.perf_counter() - t0) * 1e3


# warm chain on the moving cloud: the input of step k is the Morton order of step k-1, displaced
bvh = ibvh.BVH(vols)
times, disorder, moved = [], [], []
for k in range(args.steps + 3):
    prev = bvh.leaves.volume.contiguous()                      # last step's leaves, in Morton order (build.jl:118-121)
    prev[:, :3] += (torch.rand((n, 3), generator=g, device="cuda") * 2 - 1) * step
    old_keys = bvh.leaves.morton_device.clone()
    bvh, ms = timed_build(prev, bvh)
    if k >= 3:
        times.append(ms)
        # how sorted was the input?  keys of the NEW codes in INPUT order = new codes gathered by new index
        idx = bvh.leaves.index.long() - 1
        keys_in_input_order = torch.empty(n, dtype=torch.int64, device="cuda")
        keys_in_input_order[idx] = bvh.leaves.morton_device
        disorder.append(float((keys_in_input_order[1:] < keys_in_input_order[:-1]).float().mean()))
        moved.append(float(((keys_in_input_order >> 19) != (old_keys >> 19)).float().mean()))
lib.call("ibvh_profile_enable", 1)
prev = bvh.leaves.volume.contiguous()
prev[:, :3] += (torch.rand((n, 3), generator=g, device="cuda") * 2 - 1) * step
bvh = ibvh.BVH(prev, cache=bvh)
torch.cuda.synchronize()
k_moving = {k: round(v[0], 4) for k, v in collect_profile(lib).items()}
lib.call("ibvh_profile_enable", 0)

# the same cloud in random order
perm = torch.randperm(n, generator=g, device="cuda")
shuffled = bvh.leaves.volume.contiguous()[perm].contiguous()
b2 = ibvh.BVH(shuffled)
t_shuf = []
for k in range(args.steps + 3):
    b2, ms = timed_build(shuffled, b2)
    if k >= 3:
        t_shuf.append(ms)
lib.call("ibvh_profile_enable", 1)
b2 = ibvh.BVH(shuffled, cache=b2)
torch.cuda.synchronize()
k_shuf = {k: round(v[0], 4) for k, v in collect_profile(lib).items()}
lib.call("ibvh_profile_enable", 0)
out = {"leaves": n, "displacement_cells": args.cells, "steps": args.steps,
       "build_ms_moving_cloud_nearly_sorted_input": round(sum(times) / len(times), 4),
       "build_ms_same_cloud_shuffled_input": round(sum(t_shuf) / len(t_shuf), 4),
       "input_descents_fraction": round(sum(disorder) / len(disorder), 4),
       "leaves_changing_coarse_cell_fraction": round(sum(moved) / len(moved), 4),
       "kernels_ms_moving": k_moving, "kernels_ms_shuffled": k_shuf,
       "note": "host-synchronised builds (launch + sync overhead included in both figures alike)"}
print(json.dumps(out))
