#!/usr/bin/env python3
"""Run ON THE GPU BOX: the in-place rebuild of pre-wrapped records (`bvh = BVH(bvh.leaves; cache=bvh)`, build.jl:109-126) at 1e6 and 1e7
leaves: ms per build over ten chained rebuilds and the per-kernel events."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from bench import _dominant
for n in (1_000_000, 10_000_000):
    v = ibvh.generate_spheres(n, 42)
    st = {"b": ibvh.BVH(v)}
    ref = st["b"].leaves.buf.clone()
    def run():
        st["b"] = ibvh.BVH(st["b"].leaves, cache=st["b"])
        return st["b"]
    for _ in range(4):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    _, _, ks = _dominant(lib, torch, run)
    print(f"n={n}: in-place rebuild {ms:.3f} ms, records unchanged {bool(torch.equal(ref, st['b'].leaves.buf))}  " +
          " ".join(f"{k.replace('_kernel','')}={x:.3f}" for k, x in sorted(ks.items(), key=lambda kv: -kv[1])[:9]), flush=True)
