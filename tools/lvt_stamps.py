#!/usr/bin/env python3
"""Run ON THE GPU BOX with IBVH_LIB=variants/libibvh_stamps.so (tools/phase_stamps.sh): share of a wave's time per
section of lvt_queue_kernel's counting pass (s_memtime ticks summed over all waves of the launches).
usage: python tools/lvt_stamps.py [n]"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
b = ibvh.BVH(v)
t = ibvh.traverse(b)
t = ibvh.traverse(b, cache=t)
torch.cuda.synchronize()
L = lib.load()
L.ibvh_debug_lvt_ticks.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(8, dtype=np.uint64)
assert L.ibvh_debug_lvt_ticks(buf.ctypes.data, 1) == 0
reps = 5
for _ in range(reps):
    t = ibvh.traverse(b, cache=t)
torch.cuda.synchronize()
assert L.ibvh_debug_lvt_ticks(buf.ctypes.data, 0) == 0
names = ["prologue (load, boxes, split)", "descent to the cut level", "subtree prologue", "candidate loops", "leaf tests (drains)", "epilogue"]
tot = float(buf[:6].sum())
waves = int(buf[7])
print(f"n = {n}, contacts {t.num_contacts}, waves stamped {waves} (count + write passes of {reps} traversals)")
for k, nm in enumerate(names):
    print(f"  {nm:32s} {100.0 * float(buf[k]) / tot:5.1f} %   {float(buf[k]) / waves:9.0f} ticks / wave")
print(f"  {'total':32s}         {tot / waves:9.0f} ticks / wave (s_memtime: 100 MHz)")
