#!/usr/bin/env python3
"""Run ON THE GPU BOX: hunt for builds that take milliseconds instead of ~0.1 ms (seen once for Morton-sorted input)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from bench import collect_profile
n = 1_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
b0 = ibvh.BVH(v)
sv = b0.leaves.volume.contiguous()
for name, vols in (("uniform", v), ("sorted", sv)):
    b = None
    times = []
    for it in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b = ibvh.BVH(vols, cache=b)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    print(name, "per-build ms:", " ".join("%.2f" % t for t in times), "hint", int(b._skew[0]))
    lib.call("ibvh_profile_enable", 1)
    b = ibvh.BVH(vols, cache=b); torch.cuda.synchronize()
    p = collect_profile(lib)
    lib.call("ibvh_profile_enable", 0)
    print("   ", {k: (round(x[0] * 1e3, 1), x[1]) for k, x in p.items()})
# does a hint word pinned LATE (after the GPU has been busy) make builds slow?
b = ibvh.BVH(v)
for label, word in (("the build's own word", None), ("a word pinned just now", torch.zeros(1, dtype=torch.int32).pin_memory()),
                    ("a device word", "dev")):
    if word is not None:
        if isinstance(word, str):
            class Dev:
                def __init__(self):
                    self.t = torch.zeros(1, dtype=torch.int32, device="cuda")
                def __getitem__(self, i):
                    return 0
                def data_ptr(self):
                    return self.t.data_ptr()
            b._skew = Dev()
        else:
            b._skew = word
    times = []
    for it in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b = ibvh.BVH(v, cache=b)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    print(label, "per-build ms:", " ".join("%.2f" % t for t in times))
