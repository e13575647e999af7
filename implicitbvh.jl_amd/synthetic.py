"""Synthetic inputs of BASELINE.json's configurations, shared by bench.py, the tools and the tests.
Host-side input generation only (numpy): nothing here is on the hot path."""
import math

import numpy as np


def sphere_radius_law(n, k=8.0):
    """r0 of the config-2 law r = r0 (0.5 + 0.5 u): r0 = 0.5 (3 k / (4 pi n))^(1/3), k = 8 -> ~1.76 contacts per leaf."""
    return 0.5 * (3 * k / (4 * math.pi * n)) ** (1 / 3)


def torus_mesh(u=1898, v=1897):
    """Deterministic surrogate for xyzrgb_dragon.obj (absent from the reference repo, benchmark/README.md:3):
    a displaced torus tessellation with 2*u*v ~ 7.2 M triangles, i.e. a 2-manifold leaf distribution.
    Returns (2*u*v, 9) float32: p1 p2 p3 per triangle."""
    a = (np.arange(u, dtype=np.float64) / u * 2 * np.pi)[:, None]
    b = (np.arange(v, dtype=np.float64) / v * 2 * np.pi)[None, :]
    r = 0.35 + 0.05 * np.sin(7 * a) * np.cos(5 * b)
    x = ((1.0 + r * np.cos(b)) * np.cos(a)).astype(np.float32)
    y = ((1.0 + r * np.cos(b)) * np.sin(a)).astype(np.float32)
    z = (r * np.sin(b) + 0 * a).astype(np.float32)
    p = np.stack([x, y, z], axis=-1)
    p00, p10 = p, np.roll(p, -1, axis=0)
    p01, p11 = np.roll(p, -1, axis=1), np.roll(np.roll(p, -1, axis=0), -1, axis=1)
    t1 = np.stack([p00, p10, p11], axis=2).reshape(-1, 3, 3)
    t2 = np.stack([p00, p11, p01], axis=2).reshape(-1, 3, 3)
    return np.concatenate([t1, t2]).reshape(-1, 9)


def random_rays(num_rays, lo, hi, seed=43):
    """benchmark/bvh_rays.jl:36-38: points and directions i.i.d. uniform [0, 1), points scaled into the AABB [lo, hi].
    Returns (points, directions), each (num_rays, 3) float32 (row i = ray i = column i of the reference's (3, N))."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    p = (lo + (hi - lo) * rng.random((num_rays, 3))).astype(np.float32)
    d = rng.random((num_rays, 3)).astype(np.float32)
    return p, d
