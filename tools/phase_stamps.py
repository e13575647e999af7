#!/usr/bin/env python3
"""Run ON THE GPU BOX with IBVH_LIB=tools/libibvh_stamps.so: average cycles per phase of the sort's partition and
finish kernels (diagnostic build, tools/phase_stamps.sh)."""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
b = None
for _ in range(3):
    b = ibvh.BVH(v, cache=b)
torch.cuda.synchronize()
buf = np.zeros((2, 12, 4096), dtype=np.uint64)
L = lib.load()
L.ibvh_debug_stamps.argtypes = [C.c_void_p]
assert L.ibvh_debug_stamps(buf.ctypes.data) == 0
names = [["zero+keys", "rank", "wave prefix", "2 scans", "tile_hist column + pos", "stage records", "write out"],
         ["start (digit totals in front)", "key loads", "lds passes", "copy out"]]
for k, nm in enumerate(names):
    s = buf[k].astype(np.int64)
    used = s[0] != 0
    print("kernel", ["partition", "finish"][k], "blocks sampled", int(used.sum()))
    for p, label in enumerate(nm):
        ok = used & (s[p + 1] != 0)
        d = (s[p + 1] - s[p])[ok]
        print(f"   {label:28s} {d.mean():10.0f} ticks  (median {np.median(d):8.0f})")
    ok = used & (s[len(nm)] != 0)
    print(f"   {'total':28s} {(s[len(nm)] - s[0])[ok].mean():10.0f}")
    if k == 1:
        for p, label in zip(range(5, 9), ["pass 1: zero counters", "pass 1: rank", "pass 1: wave prefix + scan", "pass 1: scatter to LDS"]):
            ok = used & (s[p + 1] != 0) & (s[p] != 0)
            d = (s[p + 1] - s[p])[ok]
            if len(d):
                print(f"   {label:28s} {d.mean():10.0f} ticks  (median {np.median(d):8.0f})")
