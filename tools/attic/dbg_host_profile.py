#!/usr/bin/env python3
"""Run ON THE GPU BOX: where the HOST spends a bench step (cProfile over 2000 steps with the count read every step):
the GPU idles from the moment the host sees the count until the next build's first kernel arrives."""
import cProfile, math, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import implicitbvh_amd as ibvh
n = 1_000_000
v = ibvh.generate_spheres(n, 42, r0=0.5 * (3 * 8 / (4 * math.pi * n)) ** (1 / 3))
st = (None, None)


def steps(k, read):
    global st
    for _ in range(k):
        b = ibvh.BVH(v, cache=st[0])
        t = ibvh.traverse(b, cache=st[1])
        if read:
            t.num_contacts
        st = (b, t)


steps(50, True)
for read in (False, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps(500, read)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("read" if read else "enqueue only", "issue %.1f us/step, done %.1f us/step" % ((t1 - t0) / 500 * 1e6, (t2 - t0) / 500 * 1e6))
pr = cProfile.Profile()
pr.enable()
steps(2000, True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
