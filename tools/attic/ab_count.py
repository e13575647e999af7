#!/usr/bin/env python3
"""Run ON THE GPU BOX: the LVT self-traversal of config 2's cloud (1e6 and 1e7 leaves) with the dual descent (lvt_dual=1) and
with round 3's lvt_queue_kernel (lvt_dual=0): time per traversal (count + scan + write, cache reused), per-kernel averages
from the library's event timers, and the two contact lists compared byte for byte.
usage: [IBVH_LIB=variants/libibvh_TAG.so] python tools/ab_count.py [sizes...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from implicitbvh_amd.synthetic import sphere_radius_law

sizes = [int(float(x)) for x in sys.argv[1:]] or [1_000_000, 10_000_000]
L = lib.load()


def kernels():
    cnt = C.c_int64()
    lib.call("ibvh_profile_count", C.byref(cnt))
    out = {}
    for i in range(cnt.value):
        name, ms = C.c_char_p(), C.c_float()
        lib.call("ibvh_profile_get", i, C.byref(name), C.byref(ms))
        k = name.value.decode().strip("() ").split("<")[0] + ("_w" if "MODE, true" in name.value.decode() else "")
        t, c = out.get(k, (0.0, 0))
        out[k] = (t + ms.value, c + 1)
    return {k: round(1e3 * t / c, 1) for k, (t, c) in out.items() if k.startswith("lvt")}


for n in sizes:
    v = ibvh.generate_spheres(n, 42, r0=sphere_radius_law(n))
    b = ibvh.BVH(v)
    ref = None
    for dual in (1, 0, 1):
        lib.set_tuning("lvt_dual", dual)
        t = ibvh.traverse(b)
        for _ in range(5):
            t = ibvh.traverse(b, cache=t)
            t.num_contacts
        torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            t = ibvh.traverse(b, cache=t)
            t.num_contacts
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        lib.call("ibvh_profile_enable", 1)
        for _ in range(5):
            t = ibvh.traverse(b, cache=t)
            t.num_contacts
        torch.cuda.synchronize()
        ks = kernels()
        lib.call("ibvh_profile_enable", 0)
        c = t.contacts.clone()
        if ref is None:
            ref = c
        same = ref.shape == c.shape and bool(torch.equal(ref, c))
        print(f"n={n} dual={dual} traverse {ms:.4f} ms contacts {t.num_contacts} kernels_us {ks} identical_to_first {same}", flush=True)
