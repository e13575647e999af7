#!/usr/bin/env python3
"""Run ON THE GPU BOX: the build's sort under different geometries / the LDS-resident finish (ibvh_set_tuning knobs), per-kernel
times of a cached rebuild, the sorted records compared byte for byte with the default setting's.
usage: python tools/ab_sort.py [n ...]   (settings: the SETTINGS list below, or IBVH_AB_SETTINGS="k=v,k=v;k=v,...")"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import implicitbvh_amd as ibvh
from implicitbvh_amd import lib
from implicitbvh_amd.synthetic import sphere_radius_law
from bench import collect_profile, MORTON_SORT_KERNELS

KNOBS = ("msd_resident_kb", "msd_bits", "msd_cap", "msd_ftpb", "msd_tile", "msd_finish_pad_kb")
SETTINGS = (os.environ["IBVH_AB_SETTINGS"].split(";") if "IBVH_AB_SETTINGS" in os.environ else None) or [
    "", "msd_resident_kb=80", "msd_resident_kb=80,msd_bits=12,msd_cap=2816,msd_ftpb=256", "msd_resident_kb=80,msd_bits=12,msd_cap=3072,msd_ftpb=256",
    "msd_resident_kb=160,msd_bits=12,msd_cap=4096,msd_ftpb=512", "msd_resident_kb=160,msd_bits=11,msd_cap=8192,msd_ftpb=512"]
sizes = [int(float(x)) for x in sys.argv[1:]] or [1_000_000, 10_000_000]
for n in sizes:
    v = ibvh.generate_spheres(n, 42, r0=sphere_radius_law(n))
    ref = None
    for setting in SETTINGS:
        for k in KNOBS:
            lib.set_tuning(k, 0)
        for kv in filter(None, setting.split(",")):
            k, _, val = kv.partition("=")
            lib.set_tuning(k, int(val))
        try:
            b, t = ibvh.BVH(v), None
            for _ in range(5):
                b = ibvh.BVH(v, cache=b)
                t = ibvh.traverse(b, cache=t)
                t.num_contacts
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                b = ibvh.BVH(v, cache=b)
                t = ibvh.traverse(b, cache=t)
                t.num_contacts
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            lib.call("ibvh_profile_enable", 1)
            for _ in range(5):
                b = ibvh.BVH(v, cache=b)
                t = ibvh.traverse(b, cache=t)
            torch.cuda.synchronize()
            prof = collect_profile(lib)
            lib.call("ibvh_profile_enable", 0)
            ks = {k: round(1e3 * t / 5, 1) for k, (t, c) in prof.items()}  # us per step
            phase = sum(t for k, (t, c) in prof.items() if k in MORTON_SORT_KERNELS) / 5  # per step (5 profiled steps)
            buf = b.leaves.buf.clone()
            if ref is None:
                ref = buf
            print(f"n={n} [{setting or 'default'}] step {ms:.4f} ms  morton+sort {phase*1e3:.1f} us = {152.0*n/(phase*1e-3)/8e12*100:.1f} %  "
                  f"{ {k: v for k, v in ks.items() if k in MORTON_SORT_KERNELS} }  same_as_default {bool(torch.equal(ref, buf))}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"n={n} [{setting}] FAILED: {e}", flush=True)
    for k in KNOBS:
        lib.set_tuning(k, 0)
