#!/usr/bin/env python3
"""Run ON THE GPU BOX: A/B of tuning-knob settings on the LVT self-traversal of config 2's cloud (and, with --dense, config 3's
mesh self-traversal and config 4's pair traversal): time per traversal (count + scan + write, cache reused), per-kernel averages
from the library's event timers, and every setting's contact list compared byte for byte with the first one's.
usage: [IBVH_LIB=variants/libibvh_TAG.so] python tools/ab_knob.py "knob=v[,knob2=w];knob=v2;..." [--dense] [sizes...]
e.g.   python tools/ab_knob.py "lvt_blocks=0;lvt_blocks=1,lvt_block_shift=9;lvt_blocks=1,lvt_block_shift=10" 1e6 1e7"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from implicitbvh_amd.synthetic import sphere_radius_law, torus_mesh

args = [a for a in sys.argv[1:] if not a.startswith("--")]
dense = "--dense" in sys.argv
settings = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(",") if kv) for s in args[0].split(";")]
sizes = [int(float(x)) for x in args[1:]] or [1_000_000, 10_000_000]
lib.load()


def kernels():
    cnt = C.c_int64()
    lib.call("ibvh_profile_count", C.byref(cnt))
    out = {}
    for i in range(cnt.value):
        name, ms = C.c_char_p(), C.c_float()
        lib.call("ibvh_profile_get", i, C.byref(name), C.byref(ms))
        nm = name.value.decode()
        k = nm.strip("() ").split("<")[0] + ("_w" if "MODE, true" in nm else "")
        t, c = out.get(k, (0.0, 0))
        out[k] = (t + ms.value, c + 1)
    return {k: round(1e3 * t / c, 1) for k, (t, c) in out.items() if k.startswith("lvt")}


def ab(label, fn):
    ref, defaults = None, {}
    for s in settings + settings[:1]:
        for k, v in s.items():
            if k not in defaults:
                cur = C.c_int32()
                lib.call("ibvh_get_tuning", k.encode(), C.byref(cur))
                defaults[k] = cur.value
            lib.set_tuning(k, v)
        t = fn(None)
        for _ in range(5):
            t = fn(t)
            t.num_contacts
        torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            t = fn(t)
            t.num_contacts
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        lib.call("ibvh_profile_enable", 1)
        for _ in range(5):
            t = fn(t)
            t.num_contacts
        torch.cuda.synchronize()
        ks = kernels()
        lib.call("ibvh_profile_enable", 0)
        c = t.contacts.clone()
        if ref is None:
            ref = c
        same = ref.shape == c.shape and bool(torch.equal(ref, c))
        print(f"{label} {s} traverse {ms:.4f} ms contacts {t.num_contacts} kernels_us {ks} identical_to_first {same}", flush=True)
        for k, v in defaults.items():
            lib.set_tuning(k, v)


for n in sizes:
    v = ibvh.generate_spheres(n, 42, r0=sphere_radius_law(n))
    b = ibvh.BVH(v)
    ab(f"self n={n}", lambda t: ibvh.traverse(b, cache=t))
    del v, b
    torch.cuda.empty_cache()
if dense:
    tris = torch.from_numpy(torus_mesh()).cuda()
    vols = ibvh.bounding_volumes_from_triangles(tris)
    b3 = ibvh.BVH(vols)
    ab("config3 self", lambda t: ibvh.traverse(b3, cache=t))
    del tris, vols, b3
    torch.cuda.empty_cache()
    n4 = 5_000_000
    r0 = sphere_radius_law(n4)
    b1 = ibvh.BVH(ibvh.generate_spheres(n4, 44, r0=r0))
    b2 = ibvh.BVH(ibvh.generate_spheres(n4, 45, origin=(0.9, 0.0, 0.0), r0=r0))
    ab("config4 pair", lambda t: ibvh.traverse(b1, b2, cache=t))
