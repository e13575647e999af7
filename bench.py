#!/usr/bin/env python3
"""bench.py — BVH build + self-traverse throughput on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic leaves that already sit in HBM:
    BVH(volumes, BBox{Float32}; cache=previous)  ->  traverse(bvh, LVTTraversal(); cache=previous)  ->  num_contacts
on `--n` BSphere{Float32} leaves (default 1e6: BASELINE.json configs[1]), UInt32 Morton, Int32
indices.  Protocol follows the reference's benchmark scripts (benchmark/bvh_build.jl:38-45,
bvh_contact.jl:40-45): warm-up, then timed repetitions on resident data with buffer reuse.  The timed
step INCLUDES the host's read of the contact count, which the reference's traverse() performs
(`@allowscalar`, lvt/traverse_single.jl:60): `value` / `ms_per_step` are that like-for-like figure;
`value_enqueue_only` is the same K steps chained without the read.

N > 1: one process per GPU (launched by torch.distributed.run, or — `python bench.py --gpus N` with no
WORLD_SIZE in the environment — started by this script itself before anything touches a GPU): the leaves
are sharded over the ranks; the build's global centre AABB is an RCCL all-reduce and the Morton sort a
distributed radix-sort exchange (implicitbvh_amd.dist); each rank then builds and self-traverses its slice
(BASELINE.json configs[4]: 1e8 leaves over 8 GPUs = 12.5 M leaves per GPU, the default for N > 1).  Weak
scaling: --n is the per-GPU leaf count, value = all ranks' leaves / max-over-ranks time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the library),
`cpu_baseline` (the CPU oracle's multi-threaded restatement, timed on this box's host cores) and — at one
GPU — `configs`: BASELINE.json configs 2 (BFS), 3 (mesh: build, self-traverse, 1e6 rays) and 4 (two 5e6
clouds: pair LVT and BFS), each with its own roofline and, for rays and pair, its own CPU baseline.
"""
import argparse
import ctypes as C
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# kernels of the Morton+sort phase (extrema -> keys -> radix passes -> sorted records), priced at 152 B/leaf
MORTON_SORT_KERNELS = ("extrema_partial_kernel", "extrema_final_kernel", "encode_kernel", "encode_hist_kernel", "hist_kernel",
                       "hist_wide_kernel", "scan_kernel", "bucket_start_kernel", "scatter_kernel", "scatter_wide_kernel",
                       "scatter_records_kernel", "bucket_sort_kernel", "gather_kernel", "scan_tiles_kernel", "plan_kernel",
                       "partition_kernel", "finish_kernel", "finish_resident_kernel", "range_kernel", "hist_level_kernel", "scan_level_kernel")
TRAVERSE_PREFIXES = ("lvt_", "scan_reduce", "scan_apply", "scan_fused", "rays_", "level_kernel", "fill_")


# Algorithmic bytes per LEAF and launch for the kernels of one step, for
# BSphere{F32} leaves / BBox{F32} nodes / U32 / I32; C = contacts per leaf.
def algorithmic_bytes(kernel, n, contacts):
    c = contacts / max(n, 1)
    table = {
        "extrema_partial_kernel": 16.0,            # read raw volumes
        "encode_kernel": 16.0 + 4.0,               # read volumes, write keys (positions are implicit)
        "encode_hist_kernel": 16.0 + 4.0,          # same, fused with the first per-tile digit histogram
        "scatter_records_kernel": 8.0 + 16.0 + 24.0,  # last pass: read (key, pos) + source volume, write the record
        "scatter_wide_kernel": 4.0 + 8.0,             # MSD partition: read keys (positions implicit), write (key, pos)
        "bucket_sort_kernel": 8.0 + 16.0 + 24.0,      # read (key, pos) + source volume, write the sorted record
        "partition_kernel": 4.0 + 16.0 + 24.0,        # MSD partition of whole records: read key + source volume, write the record
        "finish_kernel": 24.0 + 24.0,                 # per-bucket in-LDS finish: read the partitioned record, write the sorted one
        "finish_resident_kernel": 24.0 + 24.0,        # the same, records resident in LDS (cells of the 8,192-record geometry)
        "hist_kernel": 4.0,                        # read keys
        "scatter_kernel": 8.0 + 8.0,               # read + write (key, position)
        "gather_kernel": 4.0 + 4.0 + 16.0 + 24.0,  # perm + key + volume -> record
        "aggregate_kernel": 16.0 + 24.0 + 24.0,    # leaves' volumes + every node read once + written once
        "lvt_joint_kernel_count": 24.0 + 24.0 + 4.0 + 8.0 * c,  # leaves + nodes once, counts, contact cache written
        "lvt_queue_kernel_count": 24.0 + 24.0 + 4.0 + 8.0 * c,
        "lvt_dual_kernel_count": 24.0 + 24.0 + 4.0 + 8.0 * c,
        "lvt_joint_kernel_write": 8.0 + 16.0 * c,               # prefix read (2 x 4) + cached contacts read and written
        "lvt_queue_kernel_write": 8.0 + 16.0 * c,
        "lvt_dual_kernel_write": 8.0 + 16.0 * c,
        "scan_reduce_kernel": 4.0, "scan_apply_kernel": 8.0,
    }
    return table.get(kernel, 0.0) * n


def kernel_key(name):
    """'(lvt_queue_kernel<L, N, I, MODE, true, false>)' -> 'lvt_queue_kernel_write' (5th argument = WRITE)."""
    base = name.strip("() ").split("<")[0].split("::")[-1].strip()
    if base == "scatter_kernel" and "true>" in name.replace(" ", ""):
        return "scatter_records_kernel"
    if base in ("lvt_rays_kernel", "lvt_joint_kernel", "lvt_queue_kernel", "lvt_dual_kernel", "rays_subtree_kernel"):
        flat = name.replace(" ", "")
        return base + ("_write" if ("MODE,true" in flat or "I,true>" in flat or "I,true," in flat) else "_count")
    return base


def pmc_key(demangled):
    """rocprofv3's demangled kernel name -> the same key kernel_key() gives the library's launch label."""
    import re
    m = re.match(r"(?:void )?(?:\w+::)*(\w+)(<.*>)?\(", demangled)
    if not m:
        return demangled
    base, targs = m.group(1), m.group(2) or ""
    flags = re.findall(r"\b(true|false)\b", targs)
    if base == "scatter_kernel" and flags[:1] == ["true"]:
        return "scatter_records_kernel"
    if base in ("lvt_rays_kernel", "lvt_joint_kernel", "lvt_queue_kernel", "lvt_dual_kernel", "rays_subtree_kernel"):
        return base + ("_write" if flags[:1] == ["true"] else "_count")
    return base


def csrc_sha(root=ROOT):
    """Hash of everything libibvh.so is compiled from (`__graft_entry__.kernel_sources`: *.hip, *.hpp, *.inc, the
    Makefile with its flags, include/ibvh.h): the committed counter profiles are stamped with it, and a profile taken at
    another state of the kernels is not quoted (measure-or-omit)."""
    from __graft_entry__ import kernel_sources
    h = hashlib.sha256()
    for f in kernel_sources(root):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def collect_profile(lib):
    cnt = C.c_int64()
    lib.call("ibvh_profile_count", C.byref(cnt))
    out = {}
    for i in range(cnt.value):
        name, ms = C.c_char_p(), C.c_float()
        lib.call("ibvh_profile_get", i, C.byref(name), C.byref(ms))
        k = kernel_key(name.value.decode())
        tot, num = out.get(k, (0.0, 0))
        out[k] = (tot + ms.value, num + 1)
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (children of a parent that never
    touches a GPU), wait for them, pass rank 0's JSON line through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # wait for all ranks; a rank that dies takes the others with it (they would wait for it until the RCCL timeout)
    import threading
    import time as _time
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()  # our own children, by handle
            break
        _time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write("".join(c for c in chunks if c))
    sys.stdout.flush()
    raise SystemExit(1 if failed else max(abs(rc) for rc in rcs))


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs 2 (BFS), 3 and 4 at one GPU: timing, rooflines, work counters, CPU baselines
# ------------------------------------------------------------------------------------------------------------------
def _timed(torch, fn, reps):
    # two untimed calls: a `cache=` chain reaches its steady state with the SECOND call (the first one with a cache sizes the
    # scratch from the cached contact buffer — for config 3's 82 M contacts a 3.7 GB allocation, 35 ms once)
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
        _ = out.num_contacts if hasattr(out, "num_contacts") else None  # the reference's traverse() returns the count
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def _dominant(lib, torch, fn):
    """(kernel key, average launch ms, launches) of the kernel with the largest total time in one call of fn"""
    lib.call("ibvh_profile_enable", 1)
    fn()
    torch.cuda.synchronize()
    prof = collect_profile(lib)
    lib.call("ibvh_profile_enable", 0)
    if not prof:
        return None, None, {}
    dom = max(prof, key=lambda k: prof[k][0])
    return dom, prof[dom][0] / prof[dom][1], {k: round(v[0], 4) for k, v in prof.items() if v[0] > 0.005}


def _roof(kernel, avg_ms, bytes_per_launch, note):
    if not avg_ms:
        return None
    gbps = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": round(gbps, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbps / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes": note}


def _with_block_kernel(roofline, ms_by_kernel, algorithmic_bytes_):
    """Round 5: the counting pass's descent is shared per block of leaves and runs as a small kernel in FRONT of it
    (lvt_block_frontier_kernel).  `frac` stays the dominant kernel's own; frac_with_block_kernel prices the same algorithmic
    bytes against both kernels' time — the like-for-like figure against round 4's single kernel."""
    if roofline and roofline.get("kernel") == "lvt_queue_kernel_count" and ms_by_kernel.get("lvt_block_frontier_kernel"):
        both = roofline["avg_launch_ms"] + ms_by_kernel["lvt_block_frontier_kernel"]
        roofline["block_kernel_ms"] = round(ms_by_kernel["lvt_block_frontier_kernel"], 5)
        roofline["frac_with_block_kernel"] = round(algorithmic_bytes_ / (both * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)


def _work(ibvh, orc, gpu_args, ref_args, items, threads, native):
    """tests per work item of the HIP walk and of the reference's walk (instrumented oracle) on the same input"""
    g = ibvh.lvt_work_counters(*gpu_args[0], **gpu_args[1])
    nt, lt, _ = orc.lvt_test_counts(*ref_args[0], threads=threads, native=native, **ref_args[1])
    return {"hip": {"node_tests_per_item": round(g["node_tests"] / items, 2), "leaf_tests_per_item": round(g["leaf_tests"] / items, 2),
                    "touched_bytes": g["touched_bytes"]},
            "reference_walk": {"node_tests_per_item": round(nt / items, 2), "leaf_tests_per_item": round(lt / items, 2),
                               "touched_bytes": 24 * (nt + lt)},
            "tests_hip_over_reference": round((g["node_tests"] + g["leaf_tests"]) / max(nt + lt, 1), 3), "items": items}


def run_configs(args, ibvh, lib, torch, cpu):
    """cpu: None, or (oracle_lib, native library or None, thread count) for the CPU legs."""
    import numpy as np
    from implicitbvh_amd import abi
    from implicitbvh_amd.synthetic import random_rays, sphere_radius_law, torus_mesh
    out = {}
    orc, native, threads = cpu if cpu else (None, None, 0)

    # ---- config 2 with BFS (LVT is the headline) -------------------------------------------------------------
    n2 = 1_000_000
    v2 = ibvh.generate_spheres(n2, 42, r0=sphere_radius_law(n2))
    b2 = ibvh.BVH(v2)
    st = {"t": None}

    def bfs2():
        st["t"] = ibvh.traverse(b2, ibvh.BFSTraversal(), cache=st["t"])
        return st["t"]
    ms, t = _timed(torch, bfs2, 5)
    dom, avg, ks = _dominant(lib, torch, bfs2)
    cbytes = 48.0 * n2 + 8.0 * t.num_contacts
    out["config2_bfs"] = {"workload": "1e6 random BSphere{Float32} leaves, traverse(bvh, BFSTraversal(); cache)", "ms": round(ms, 4),
                          "contacts": t.num_contacts, "num_checks": t.num_checks,
                          "mcontacts_per_s": round(t.num_contacts / ms / 1e3, 1), "kernels_ms": ks,
                          "roofline": _roof("whole traversal (one level kernel launch per tree level)", ms, cbytes,
                                            "leaves 24 + nodes 24 per leaf, each once + 8 per contact (queue traffic is the implementation's)")}
    del v2, b2, t, st
    torch.cuda.empty_cache()

    # ---- config 3: mesh (IBVH_MESH=/path/to/xyzrgb_dragon.obj, else the 7.2 M-triangle torus surrogate) + 1e6 rays ----
    mesh_path = os.environ.get("IBVH_MESH", "")
    if mesh_path and os.path.exists(mesh_path):
        tris = ibvh.load_obj_triangles(mesh_path)
        mesh_name = mesh_path
        tris_host = None
    else:
        tris_host = torus_mesh()
        tris = torch.from_numpy(tris_host).cuda()
        mesh_name = "torus surrogate, 7.2 M triangles (xyzrgb_dragon.obj is not in the reference repo; set IBVH_MESH)"
    ms_vol, vols = _timed(torch, lambda: ibvh.bounding_volumes_from_triangles(tris), 3)
    n3 = int(vols.shape[0])
    s3 = {"b": None, "t": None, "r": None}

    def build3():
        s3["b"] = ibvh.BVH(vols, cache=s3["b"])
        return s3["b"]

    def self3():
        s3["t"] = ibvh.traverse(s3["b"], cache=s3["t"])
        return s3["t"]
    ms_b3, _ = _timed(torch, build3, 5)
    dom_b, avg_b, ks_b = _dominant(lib, torch, build3)
    ms_s3, t3 = _timed(torch, self3, 3)
    dom_s, avg_s, ks_s = _dominant(lib, torch, self3)
    hv = vols[:, :3]
    lo, hi = hv.min(0).values.cpu().numpy(), hv.max(0).values.cpu().numpy()
    nr = 1_000_000
    p_host, d_host = random_rays(nr, lo, hi, seed=43)
    p, d = torch.from_numpy(p_host).cuda().t(), torch.from_numpy(d_host).cuda().t()

    def rays3():
        s3["r"] = ibvh.traverse_rays(s3["b"], p, d, cache=s3["r"])
        return s3["r"]
    ms_r3, r3 = _timed(torch, rays3, 3)
    dom_r, avg_r, ks_r = _dominant(lib, torch, rays3)
    s3["rb"] = None

    def rays3_bfs():  # the reference's benchmark times both algorithms (benchmark/bvh_rays.jl:54-57)
        s3["rb"] = ibvh.traverse_rays(s3["b"], p, d, ibvh.BFSTraversal(), cache=s3["rb"])
        return s3["rb"]
    ms_rb3, rb3 = _timed(torch, rays3_bfs, 3)
    _, _, ks_rb = _dominant(lib, torch, rays3_bfs)
    c3 = {"workload": f"{mesh_name}: BSphere{{Float32}} leaves from triangles, build, self-traverse, traverse_rays with {nr} random rays "
                      "(benchmark/bvh_rays.jl:36-58)",
          "triangles": n3, "volumes_ms": round(ms_vol, 4),
          "build": {"ms": round(ms_b3, 4), "mleaves_per_s": round(n3 / ms_b3 / 1e3, 1), "kernels_ms": ks_b,
                    "roofline": _roof("whole build", ms_b3, 216.0 * n3, "216 B/leaf (SURVEY.md §8d build total)")},
          "self": {"ms": round(ms_s3, 4), "contacts": t3.num_contacts, "mcontacts_per_s": round(t3.num_contacts / ms_s3 / 1e3, 1),
                   "kernels_ms": ks_s,
                   "roofline": _roof(dom_s, avg_s, algorithmic_bytes(dom_s, n3, t3.num_contacts) or (60.0 * n3 + 8.0 * t3.num_contacts),
                                     "dominant kernel: leaves 24 + nodes 24 + counts 4 per leaf + 8 per cached contact")},
          "rays": {"ms": round(ms_r3, 4), "rays": nr, "hits": r3.num_contacts, "mrays_per_s": round(nr / ms_r3 / 1e3, 2),
                   "kernels_ms": ks_r,
                   # the default path is a chain of launches (top walk, binning, subtree walk out of LDS, placement: csrc/ibvh_lvt.hip
                   # "(3c)"), none of which is the traversal on its own: the roofline is stated for the whole call, the
                   # heaviest launch is named beside it
                   "roofline": _roof("whole traversal (rays_top_kernel + binning + rays_subtree_kernel + rays_place_kernel)", ms_r3,
                                     24.0 * nr + 4.0 * nr + 48.0 * n3 + 8.0 * r3.num_contacts,
                                     "rays 24 + counts 4 per ray, tree (leaves 24 + nodes 24 per leaf) once, 8 per hit"),
                   "dominant_kernel": {"kernel": dom_r, "avg_launch_ms": round(avg_r, 5) if avg_r else None}},
          "rays_bfs": {"ms": round(ms_rb3, 4), "rays": nr, "hits": rb3.num_contacts, "num_checks": rb3.num_checks,
                       "mrays_per_s": round(nr / ms_rb3 / 1e3, 2), "kernels_ms": ks_rb,
                       "roofline": _roof("whole traversal (one level kernel launch per tree level)", ms_rb3,
                                         24.0 * nr + 48.0 * n3 + 8.0 * rb3.num_contacts,
                                         "rays 24 per ray, tree (leaves 24 + nodes 24 per leaf) once, 8 per hit (queue traffic is the implementation's)")}}
    s3["rb"] = None
    if orc is not None:
        # CPU legs on a bounded sample: the same mesh's volumes (from the GPU: the triangle kernel is bit-exact against the
        # oracle, tests/test_gpu_parity.py), the oracle's multi-threaded build, then the two-pass LVT ray walk on the first
        # cpu_rays rays of the same ray set
        hv_all = vols.cpu().numpy()
        ob, _, tb, tt = orc.bench_build_traverse_f32(hv_all, threads, native)
        cpu_rays = 200_000
        hits, ts = orc.bench_rays_lvt(ob, p_host[:cpu_rays], d_host[:cpu_rays], threads, native)
        c3["rays"]["cpu_baseline"] = {"value": round(cpu_rays / ts / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
                                      "sample": f"first {cpu_rays} of the {nr} rays on the same {n3}-leaf tree, two-pass LVT "
                                                f"(raytrace/leaf_vs_tree), {ts * 1e3:.1f} ms, {hits} hits, one run",
                                      "gpu_over_cpu": round((nr / ms_r3 / 1e3) / (cpu_rays / ts / 1e6), 1)}
        c3["build"]["cpu_baseline"] = {"value": round(n3 / tb / 1e6, 3), "unit": "Mleaves/s", "cores": threads, "kind": "port",
                                       "sample": f"{n3} leaves, {tb * 1e3:.1f} ms, one run"}
        c3["self"]["cpu_baseline"] = {"value": round(n3 / tt / 1e6, 3), "unit": "Mleaves/s", "cores": threads, "kind": "port",
                                      "sample": f"{n3} leaves, LVT two-pass {tt * 1e3:.1f} ms, one run"}
        wr = 100_000
        c3["rays"]["work"] = _work(ibvh, orc, ((s3["b"],), {"points": p[:, :wr], "directions": d[:, :wr]}),
                                   ((ob,), {"points": p_host[:wr], "directions": d_host[:wr]}), wr, threads, native)
        c3["self"]["work"] = _work(ibvh, orc, ((s3["b"],), {}), ((ob,), {}), n3, threads, native)
        del ob, hv_all
    out["config3"] = c3
    del tris, vols, s3, t3, r3, p, d
    torch.cuda.empty_cache()

    # ---- config 4: two 5e6-leaf clouds, 10 % overlap, pair traversal (LVT and BFS) -----------------------------
    n4 = 5_000_000
    r0 = sphere_radius_law(n4)
    a = ibvh.generate_spheres(n4, 44, r0=r0)
    b = ibvh.generate_spheres(n4, 45, origin=(0.9, 0.0, 0.0), r0=r0)
    ms_b4, b1 = _timed(torch, lambda: ibvh.BVH(a), 3)
    bb = ibvh.BVH(b)
    s4 = {"t": None, "f": None}

    def pair4():
        s4["t"] = ibvh.traverse(b1, bb, cache=s4["t"])
        return s4["t"]

    def pair4_bfs():
        s4["f"] = ibvh.traverse(b1, bb, ibvh.BFSTraversal(), cache=s4["f"])
        return s4["f"]
    ms_p4, t4 = _timed(torch, pair4, 5)
    dom_p, avg_p, ks_p = _dominant(lib, torch, pair4)
    ms_f4, f4 = _timed(torch, pair4_bfs, 3)
    _, _, ks_f = _dominant(lib, torch, pair4_bfs)
    pair_bytes = 24.0 * n4 + 48.0 * n4 + 4.0 * n4 + 8.0 * t4.num_contacts
    c4 = {"workload": "two clouds of 5e6 random BSphere{Float32} leaves, 10 % overlap in x, traverse(bvh1, bvh2) "
                      "(benchmark/bvh_contact_pair.jl:38-46)",
          "leaves_each": n4, "build_ms_each": round(ms_b4, 4),
          "pair_lvt": {"ms": round(ms_p4, 4), "contacts": t4.num_contacts, "mcontacts_per_s": round(t4.num_contacts / ms_p4 / 1e3, 1),
                       "kernels_ms": ks_p,
                       "roofline": _roof(dom_p, avg_p, pair_bytes,
                                         "counting pass: driving leaves 24 + walked tree (leaves 24 + nodes 24) + counts 4 per leaf + 8 per cached contact")},
          "pair_bfs": {"ms": round(ms_f4, 4), "contacts": f4.num_contacts, "num_checks": f4.num_checks, "kernels_ms": ks_f,
                       "roofline": _roof("whole traversal", ms_f4, 2 * 48.0 * n4 + 8.0 * f4.num_contacts,
                                         "both trees (leaves 24 + nodes 24 per leaf) once + 8 per contact")}}
    if orc is not None:
        ha, hb = a.cpu().numpy(), b.cpu().numpy()
        oa, _, tba, _ = orc.bench_build_traverse_f32(ha, threads, native)
        ob2, _, _, _ = orc.bench_build_traverse_f32(hb, threads, native)
        best = None
        for _ in range(2):
            nc, ts = orc.bench_pair_lvt(oa, ob2, threads, native)
            best = ts if best is None or ts < best else best
        c4["pair_lvt"]["cpu_baseline"] = {"value": round(nc / best / 1e6, 3), "unit": "Mcontacts/s", "cores": threads, "kind": "port",
                                          "ms": round(best * 1e3, 2),
                                          "sample": f"the same two {n4}-leaf clouds, two-pass LVT pair walk (lvt/traverse_pair.jl), "
                                                    f"{nc} contacts, best of 2 runs",
                                          "contacts_match_gpu": nc == t4.num_contacts,
                                          "gpu_over_cpu": round(best * 1e3 / ms_p4, 1)}
        c4["pair_bfs"]["cpu_baseline"] = c4["pair_lvt"]["cpu_baseline"]["value"]
        c4["pair_lvt"]["work"] = _work(ibvh, orc, ((b1, bb), {}), ((oa, ob2), {}), n4, threads, native)
        del oa, ob2, ha, hb
    out["config4"] = c4
    del a, b, b1, bb, s4, t4, f4
    torch.cuda.empty_cache()

    # ---- Float64 leaves (config 1's types at config 2's size): BSphere{Float64} leaves under the default BBox{Float32} nodes
    # (build.jl:200) and under BBox{Float64} nodes — the instantiations that miss the bench types' occupancy ------------------
    n6 = 1_000_000
    v64 = ibvh.generate_spheres(n6, 42, r0=sphere_radius_law(n6)).double()
    f64 = {}
    for name, nt in (("bbox_f32_nodes", None), ("bbox_f64_nodes", ibvh.BBox(torch.float64))):
        s6 = {"b": None, "t": None}

        def step6():
            s6["b"] = ibvh.BVH(v64, nt, cache=s6["b"])
            s6["t"] = ibvh.traverse(s6["b"], cache=s6["t"])
            return s6["t"]
        ms6, t6 = _timed(torch, step6, 10)
        _, _, ks6 = _dominant(lib, torch, step6)
        f64[name] = {"ms_per_step": round(ms6, 4), "mleaves_per_s": round(n6 / ms6 / 1e3, 1), "contacts": t6.num_contacts, "kernels_ms": ks6}
        del s6, t6
    out["config2_f64_leaves"] = {"workload": f"{n6} BSphere{{Float64}} leaves (config-2 law, the Float32 cloud widened), build + LVT self-traverse + count read",
                                 **f64}
    del v64
    torch.cuda.empty_cache()

    # ---- skewed input: the same 1e6 leaves uniform and drawn from 8 tight Gaussian clusters (sigma 0.004 of
    # the box), build only, `cache=` chains — the chain's hint sends the clustered one through equalised sort cells
    # (DESIGN.md §3 "Equalised cells") —, ten chained rebuilds per figure -------------------------------------------------
    ns = 1_000_000
    g = torch.Generator(device="cuda").manual_seed(7)
    centres = torch.rand((8, 3), generator=g, device="cuda")
    clustered = torch.empty((ns, 4), dtype=torch.float32, device="cuda")
    clustered[:, :3] = centres[torch.randint(0, 8, (ns,), generator=g, device="cuda")] + 0.004 * torch.randn((ns, 3), generator=g, device="cuda")
    clustered[:, 3] = 1e-4
    uniform = ibvh.generate_spheres(ns, 42)
    skew = {}
    for name, vv in (("uniform", uniform), ("clusters8", clustered)):
        sk = {"b": None}

        def build_sk():
            sk["b"] = ibvh.BVH(vv, cache=sk["b"])
            return sk["b"]
        for _ in range(4):
            build_sk()
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            build_sk()
        torch.cuda.synchronize()
        skew[name] = {"ms": round((time.perf_counter() - t0) / 10 * 1e3, 4), "equalised_cells": int(sk["b"]._fast[1].sort_equalize) if sk["b"]._fast else None}
        del sk
    skew["clusters_over_uniform"] = round(skew["clusters8"]["ms"] / skew["uniform"]["ms"], 3)
    out["skew_1e6_build"] = skew
    del clustered, uniform
    torch.cuda.empty_cache()

    # ---- time stepping: the reference's literal loop (build.jl:109-126, README.md:84-95) on a moving cloud -----------
    out["timestep"] = {str(nn): run_timestep(nn, ibvh, lib, torch, cpu) for nn in (1_000_000, 10_000_000)}
    out["readme_250k"] = run_readme_250k(ibvh, lib, torch, cpu)
    return out


README_TRIANGLES = 249_882  # xyzrgb_dragon.obj (README.md:228-231, benchmark/bvh_contact.jl:32)
README_RAYS = 100_000       # benchmark/bvh_rays.jl:36
# the reference's only published numbers (README.md:229-231), as CONTEXT: other hardware, the real mesh
README_PUBLISHED = {"source": "/root/reference/README.md:229-231 (xyzrgb_dragon.obj, 249,882 triangles; 100,000 rays)",
                    "a100_ms": {"build": 0.40958, "traverse": 1.14, "traverse_rays": 2.00},
                    "m3max_1thread_ms": {"build": 7.11, "traverse": 67.14, "traverse_rays": 369.7},
                    "m3max_4threads_ms": {"build": 2.631, "traverse": 19.7, "traverse_rays": 113.8}}


def readme_mesh_volumes(ibvh, torch):
    """The published workload's input: IBVH_MESH (the real xyzrgb_dragon.obj) when present and of that size, else the torus
    generator cut to exactly 249,882 triangles; -> (BSphere{Float32} volumes on the GPU, name)."""
    from implicitbvh_amd.synthetic import torus_mesh
    mesh_path = os.environ.get("IBVH_MESH", "")
    if mesh_path and os.path.exists(mesh_path):
        tris = ibvh.load_obj_triangles(mesh_path)
        if int(tris.shape[0]) == README_TRIANGLES:
            return ibvh.bounding_volumes_from_triangles(tris), mesh_path
    tris = torch.from_numpy(torus_mesh(354, 353)[:README_TRIANGLES].copy()).cuda()
    return ibvh.bounding_volumes_from_triangles(tris), "torus surrogate cut to 249,882 triangles (xyzrgb_dragon.obj is not in the reference repo; set IBVH_MESH)"


def run_readme_250k(ibvh, lib, torch, cpu):
    """The reference's only PUBLISHED workload (README.md:226-231; benchmark/bvh_build.jl, bvh_contact.jl, bvh_rays.jl): 249,882
    triangle spheres, BBox{Float32} nodes, UInt32 / Int32: build (cache= chain, as :38-45), LVT self-traverse on the built BVH,
    traverse_rays with 100,000 random rays — at this size a step is a chain of dependent launches, not bandwidth."""
    from implicitbvh_amd.synthetic import random_rays
    orc, native, threads = cpu if cpu else (None, None, 0)
    vols, name = readme_mesh_volumes(ibvh, torch)
    n = int(vols.shape[0])
    st = {"b": None, "t": None, "r": None}

    def build():
        st["b"] = ibvh.BVH(vols, cache=st["b"])
        return st["b"]

    def self_():
        st["t"] = ibvh.traverse(st["b"], cache=st["t"])
        return st["t"]
    ms_b, _ = _timed(torch, build, 50)
    _, _, ks_b = _dominant(lib, torch, build)
    ms_s, t = _timed(torch, self_, 50)
    dom_s, avg_s, ks_s = _dominant(lib, torch, self_)
    hv = vols[:, :3]
    lo, hi = hv.min(0).values.cpu().numpy(), hv.max(0).values.cpu().numpy()
    p_host, d_host = random_rays(README_RAYS, lo, hi, seed=43)
    p, d = torch.from_numpy(p_host).cuda().t(), torch.from_numpy(d_host).cuda().t()

    def rays():
        st["r"] = ibvh.traverse_rays(st["b"], p, d, cache=st["r"])
        return st["r"]
    ms_r, r = _timed(torch, rays, 30)
    _, _, ks_r = _dominant(lib, torch, rays)

    def step():  # build + self-traverse + the host's read of the count, chained: what a simulation step costs
        build()
        return self_().num_contacts
    ms_step, _ = _timed(torch, step, 100)
    res = {"workload": f"{name}: BSphere{{Float32}} leaves, BBox{{Float32}} nodes, UInt32, Int32; BVH(...; cache), traverse(bvh; cache), "
                       f"traverse_rays with {README_RAYS} random rays (README.md:226-231, benchmark/bvh_contact.jl:21-45, bvh_rays.jl:36-58)",
           "leaves": n,
           "build": {"ms": round(ms_b, 4), "launches": len(ks_b), "kernels_ms": ks_b,
                     "frac": _roof("whole build", ms_b, 216.0 * n, "216 B/leaf")["frac"]},
           "self": {"ms": round(ms_s, 4), "contacts": t.num_contacts, "launches": len(ks_s), "kernels_ms": ks_s,
                    "frac": _roof("whole traversal", ms_s, 60.0 * n + 8.0 * t.num_contacts, "60 B/leaf + 8 per contact")["frac"]},
           "rays": {"ms": round(ms_r, 4), "hits": r.num_contacts, "launches": len(ks_r), "kernels_ms": ks_r,
                    "frac": _roof("whole traversal", ms_r, 28.0 * README_RAYS + 48.0 * n + 8.0 * r.num_contacts,
                                  "rays 24 + counts 4 per ray, tree 48 per leaf once, 8 per hit")["frac"]},
           "build_plus_self_step_ms": round(ms_step, 4),
           "published_for_context": README_PUBLISHED}
    if orc is not None:
        hv_all = vols.cpu().numpy()
        best = None
        for _ in range(3):
            ob, cc, tb, tt = orc.bench_build_traverse_f32(hv_all, threads, native)
            best = (tb, tt) if best is None or tb + tt < sum(best) else best
        hits, ts = orc.bench_rays_lvt(ob, p_host, d_host, threads, native)
        res["cpu_baseline"] = {"kind": "port", "cores": threads, "build_ms": round(best[0] * 1e3, 3), "self_ms": round(best[1] * 1e3, 3),
                               "rays_ms": round(ts * 1e3, 3), "contacts_match_gpu": len(cc) == t.num_contacts,
                               "hits_match_gpu": hits == r.num_contacts,
                               "sample": f"the same {n} leaves and {README_RAYS} rays, oracle build + two-pass LVT (best of 3) and LVT ray walk (one run)"}
    return res


def run_timestep(n, ibvh, lib, torch, cpu, steps=40, cells=1.0):
    """`bvh = BVH(bvh.leaves, N; cache=bvh)` IN PLACE — pre-wrapped records with user indices, the record array is input and
    output — after every leaf was moved by at most `cells` cells of the 1024^3 Morton grid, then
    `traversal = traverse(bvh; cache=traversal)` and the host's read of the contact count: the input of every build is the
    previous step's Morton order, displaced (nearly sorted).  The moves themselves (one elementwise kernel on the strided
    volume view + torch's random numbers) are timed separately and subtracted."""
    import numpy as np
    from implicitbvh_amd.synthetic import sphere_radius_law
    orc, native, threads = cpu if cpu else (None, None, 0)
    r0 = sphere_radius_law(n)
    vols = ibvh.generate_spheres(n, 47, r0=r0)
    user = torch.arange(n, 0, -1, dtype=torch.int32, device="cuda")  # user indices: reversed numbering, kept by every rebuild
    bv = ibvh.BoundingVolumes.wrap(vols, user)
    del vols
    g = torch.Generator(device="cuda").manual_seed(11)
    step = cells / 1024.0
    st = {"b": ibvh.BVH(bv), "t": None}

    def move():
        st["b"].leaves.volume[:, :3] += (torch.rand((n, 3), generator=g, device="cuda") * 2 - 1) * step

    def one():
        move()
        st["b"] = ibvh.BVH(st["b"].leaves, cache=st["b"])
        st["t"] = ibvh.traverse(st["b"], cache=st["t"])
        return st["t"].num_contacts
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        contacts = one()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        move()
    torch.cuda.synchronize()
    t_move = time.perf_counter() - t0
    one()
    ms = (t_all - t_move) / steps * 1e3
    # how sorted is the input of a build?  (keys of the new codes in INPUT order = the previous Morton order)
    move()
    old_idx = st["b"].leaves.index.clone()
    st["b"] = ibvh.BVH(st["b"].leaves, cache=st["b"])
    pos_of = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    pos_of[old_idx.long()] = torch.arange(n, device="cuda")  # input position of every user index
    keys_in = torch.empty(n, dtype=torch.int64, device="cuda")
    keys_in[pos_of[st["b"].leaves.index.long()]] = st["b"].leaves.morton_device
    descents = float((keys_in[1:] < keys_in[:-1]).float().mean())
    assert bool((st["b"].leaves.index.sort().values == torch.arange(1, n + 1, dtype=torch.int32, device="cuda")).all())  # indices kept
    dom, avg, ks = _dominant(lib, torch, one)
    ms_sort = sum(v for k, v in ks.items() if k in MORTON_SORT_KERNELS)
    bytes_step = 216.0 * n + 60.0 * n + 8.0 * contacts
    res = {"workload": f"{n} BSphere{{Float32}} leaves (config-2 law), pre-wrapped with user indices, moved <= {cells} cell per step, "
                       "bvh = BVH(bvh.leaves; cache=bvh) in place + traverse(bvh; cache=traversal) + read of the count "
                       "(build.jl:109-126, README.md:84-95)",
           "ms_per_step": round(ms, 4), "mleaves_per_s": round(n / ms / 1e3, 1), "contacts": contacts,
           "move_ms_per_step_subtracted": round(t_move / steps * 1e3, 4),
           "input_descents_fraction": round(descents, 4), "kernels_ms": ks,
           # priced at the chain's OWN input: pre-wrapped 24-byte records instead of 16-byte volumes cost extrema and encode 8 B/leaf
           # more each (SURVEY.md §8d's 152 B/leaf is for raw volumes): 168 B/leaf; frac_at_152 is the raw-volume pricing
           "morton_sort_phase": {"ms": round(ms_sort, 4), "bytes_per_leaf": 168,
                                 "frac": round(168.0 * n / (ms_sort * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_sort else None,
                                 "frac_at_152": round(152.0 * n / (ms_sort * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_sort else None},
           "roofline": _roof("whole step (build + LVT self-traverse)", ms, bytes_step,
                             "build 216 B/leaf + traverse 60 B/leaf + 8 per contact (SURVEY.md §8d)")}
    if n <= 2_000_000:
        res["roofline"]["cache_residency"] = "infinity-cache-resident: the step's working set (~130 MB) fits the 256 MB Infinity Cache"
    if orc is not None:
        # the SAME step on both sides: move the chain once more, hand the moved volumes (the build's input, in the chain's nearly
        # sorted order) to the oracle, then let the GPU take that very step and compare the contact counts
        move()
        hv = st["b"].leaves.volume.contiguous().cpu().numpy()
        st["b"] = ibvh.BVH(st["b"].leaves, cache=st["b"])
        st["t"] = ibvh.traverse(st["b"], cache=st["t"])
        gpu_contacts = int(st["t"].num_contacts)
        _, cc, tb, tt = orc.bench_build_traverse_f32(hv, threads, native)
        res["cpu_baseline"] = {"value": round(n / (tb + tt) / 1e6, 3), "unit": "Mleaves/s", "cores": threads, "threads_used": threads,
                               "kind": "port", "build_ms": round(tb * 1e3, 2), "traverse_ms": round(tt * 1e3, 2),
                               "contacts_match_gpu": len(cc) == gpu_contacts,
                               "sample": f"the {n} leaves of ONE step of the chain (the moved volumes the GPU then builds from, nearly sorted order): "
                                         f"oracle build + two-pass LVT, {len(cc)} contacts (GPU, same step: {gpu_contacts}), one run "
                                         "(the oracle wraps anew: same work, indices 1..n)"}
        res["gpu_over_cpu"] = round((n / ms / 1e3) / res["cpu_baseline"]["value"], 1)
    del st, bv
    torch.cuda.empty_cache()
    return res


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _compact_roof(r):
    if not isinstance(r, dict):
        return None
    out = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "block_kernel_ms", "frac_with_block_kernel"))
    out["traffic"] = r.get("traffic")
    if "kernel" in out and len(out["kernel"]) > 40:
        out["kernel"] = out["kernel"][:37] + "..."
    if isinstance(r.get("cache_residency"), str):
        out["cache"] = "infinity-cache" if r["cache_residency"].startswith("infinity") else "hbm"
    if isinstance(r.get("morton_sort_phase"), dict):
        out["morton_sort_phase"] = _pick(r["morton_sort_phase"], ("ms", "frac", "bytes_per_leaf"))
    if isinstance(r.get("issue"), dict):
        out["issue"] = _pick(r["issue"], ("wave_instructions_per_launch", "valu_busy_frac", "salu_busy_frac"))
    return out


def _compact_node(d):
    """One config entry -> {ms, roofline.frac, ...}; dicts without a time of their own are walked."""
    if not isinstance(d, dict):
        return None
    ms = d.get("ms", d.get("ms_per_step"))
    if ms is not None:
        out = {"ms": ms}
        if isinstance(d.get("roofline"), dict):
            out["frac"] = d["roofline"].get("frac")
        if isinstance(d.get("morton_sort_phase"), dict):
            out["morton_sort_frac"] = d["morton_sort_phase"].get("frac")
            out["morton_sort_ms"] = d["morton_sort_phase"].get("ms")
        cb = d.get("cpu_baseline")
        if isinstance(cb, dict):
            out["cpu"] = cb.get("value")
            if "contacts_match_gpu" in cb:
                out["cpu_contacts_match"] = cb["contacts_match_gpu"]
        if d.get("gpu_over_cpu") is not None:
            out["gpu_over_cpu"] = d["gpu_over_cpu"]
        elif isinstance(cb, dict) and cb.get("gpu_over_cpu") is not None:
            out["gpu_over_cpu"] = cb["gpu_over_cpu"]
        return out
    out = {}
    for k, v in d.items():
        c = _compact_node(v)
        if c:
            out[k] = c
    return out or None


def compact_line(line, detail_path):
    """The ONE line the driver records (<= 4 KB): the contract's fields, the north-star size and one {ms, frac} per config;
    everything else is in `detail_path` (a 17 KB line lost its head in the driver's tail)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "value_enqueue_only", "ms_per_step_enqueue_only", "mcontacts_per_s", "profiled_ms_per_step")
    out = {k: line[k] for k in keep if k in line}
    cfg = line.get("config") or {}
    out["config"] = _pick(cfg, ("workload", "leaves_per_gpu", "leaves_total", "contacts_total", "parallelism"))
    out["roofline"] = _compact_roof(line.get("roofline"))
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "host_cores", "contacts_match_gpu", "gpu_over_cpu", "build_ms", "traverse_ms"))
        c["sample"] = (cb.get("sample") or "")[:160]
        out["cpu_baseline"] = c
    else:
        out["cpu_baseline"] = None
    ns = line.get("north_star_1e7")
    if isinstance(ns, dict):
        o = _pick(ns, ("leaves", "ms_per_step", "value", "unit", "contacts", "build_ms", "gpu_over_cpu"))
        o["roofline"] = _compact_roof(ns.get("roofline"))
        o["morton_sort_phase"] = _pick(ns.get("morton_sort_phase") or {}, ("ms", "frac", "bytes_per_leaf"))
        if isinstance(ns.get("cpu_baseline"), dict):
            o["cpu_baseline"] = _pick(ns["cpu_baseline"], ("value", "unit", "cores", "kind"))
        out["north_star_1e7"] = o
    if line.get("configs"):
        out["configs"] = _compact_node(line["configs"])
        sk = line["configs"].get("skew_1e6_build")
        if isinstance(sk, dict) and out["configs"] is not None:
            out["configs"]["skew_1e6_build"] = {"uniform_ms": sk["uniform"]["ms"], "clusters8_ms": sk["clusters8"]["ms"], "ratio": sk["clusters_over_uniform"]}
        rd = line["configs"].get("readme_250k")
        if isinstance(rd, dict) and out["configs"] is not None:  # the reference's published workload (README.md:226-231)
            c = {k: {"ms": rd[k]["ms"], "frac": rd[k]["frac"]} for k in ("build", "self", "rays")}
            c["build_plus_self_step_ms"] = rd["build_plus_self_step_ms"]
            c["leaves"] = rd["leaves"]
            if isinstance(rd.get("cpu_baseline"), dict):
                c["cpu_ms"] = _pick(rd["cpu_baseline"], ("build_ms", "self_ms", "rays_ms", "cores", "contacts_match_gpu", "hits_match_gpu"))
            c["published_a100_ms"] = rd["published_for_context"]["a100_ms"]
            out["configs"]["readme_250k"] = c
    if isinstance(line.get("exchange"), dict):
        out["exchange"] = _pick(line["exchange"], ("max_exchange_ms", "bytes_sent_max", "xgmi_frac_per_link"))
    if isinstance(line.get("dist"), dict):
        out["dist"] = _pick(line["dist"], ("cross_ms", "contacts_cross_total", "contacts_total_with_cross", "phases_ms_max_over_ranks",
                                           "cross_bytes_received_max", "self_check"))
    if isinstance(line.get("work"), dict):
        out["work"] = _pick(line["work"], ("tests_hip_over_reference",))
    out["detail"] = detail_path
    return out

PROFILE_ROUND = "r06"


def _host_cpu():
    """physical cores, hardware threads and model name of the host (Linux /proc/cpuinfo)"""
    cores, model, phys, core = set(), "", None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and not model:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return {"cores": len(cores) or None, "threads": os.cpu_count(), "model": model or None}


def _attach_counters(roofline, tag, avg_s):
    """roofline.traffic / roofline.issue from the committed counter profiles of the dominant kernel — only while the kernel
    sources still hash to the state the profile was taken at."""
    sha = csrc_sha()
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_fetch_write_{tag}.json")
        if not os.path.exists(path):
            continue
        try:
            doc = json.load(open(path))
            if doc.get("csrc_sha") == sha:
                for name, v in doc["kernels"].items():
                    if pmc_key(name) == roofline["kernel"] and v["launches"] >= 5:
                        roofline["traffic"] = int((2 * v["FETCH_SIZE_KiB_avg"] + v["WRITE_SIZE_KiB_avg"]) * 1024)
                        roofline["traffic_source"] = (f"profiles/{rnd}_pmc_fetch_write_{tag}.json at csrc {sha} (2*FETCH_SIZE + WRITE_SIZE: "
                                                      "L2-miss bytes; Infinity-Cache hits included)")
            else:
                roofline["traffic_note"] = f"profiles/{rnd}_pmc_fetch_write_{tag}.json was taken at csrc {doc.get('csrc_sha')}, this is {sha}: omitted"
        except Exception:
            roofline["traffic_note"] = "no counter profile for this state of the kernels"
        break
    else:
        roofline["traffic_note"] = "no counter profile for this state of the kernels"
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_sq_counters_{tag}.json")
        if not os.path.exists(path):
            continue
        try:
            doc = json.load(open(path))
            sq = doc["kernels"].get(roofline["kernel"]) if doc.get("csrc_sha") == sha else None
            if sq:
                instr = sum(sq.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM",
                                                     "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
                peak = 256 * 4 * 2.4e9
                roofline["issue"] = {"wave_instructions_per_launch": int(instr), "achieved_Ginstr_per_s": round(instr / avg_s / 1e9, 1),
                                     "peak_Ginstr_per_s": round(peak / 1e9, 1), "frac": round(instr / avg_s / peak, 4),
                                     "valu_busy_frac": sq.get("valu_busy_frac"), "salu_busy_frac": sq.get("salu_busy_frac"),
                                     "source": f"profiles/{rnd}_sq_counters_{tag}.json at csrc {sha}"}
        except Exception:
            pass
        break


def emulate_config5(args):
    """configs[4] shaped for its first real node: P virtual ranks on ONE GPU run the library's distributed driver (same kernels,
    same sequence of collectives, collectives served in process).  Per rank: leaves sent / received, slice size, build and
    traverse time (the ranks share the GPU and the Python interpreter: these are serialised times of one rank's work, NOT
    a scaling measurement), contacts.  Marked emulated: nothing here says anything about xGMI."""
    import threading
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import implicitbvh_amd as ibvh
    from implicitbvh_amd import dist as ibd
    from virtual_ranks import run_virtual_ranks
    P = args.virtual_ranks
    n = args.n or 12_500_000
    r0 = 0.5 * (3 * 8 / (4 * math.pi * n * P)) ** (1 / 3)
    lock = threading.Lock()

    def fn(comm):
        vols = ibvh.generate_spheres(n, args.seed, first_index=comm.rank * n, r0=r0)
        builder = ibd.DistributedBuilder(comm)
        bvh = builder.build(vols)
        trav = ibvh.traverse(bvh)
        _ = trav.num_contacts
        torch.cuda.synchronize()
        # the build needs the other ranks for its collectives: timed as a JOINT rebuild of all ranks; the traversal of one
        # slice is timed alone on the GPU
        t0 = time.perf_counter()
        bvh = builder.build(vols, cache=bvh)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        with lock:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _k in range(3):
                trav = ibvh.traverse(bvh, cache=trav)
                _ = trav.num_contacts
            torch.cuda.synchronize()
            t_trav = (time.perf_counter() - t0) / 3
        last = builder.last
        rb = last["record_bytes"]
        # cross-shard completion (ibvh_dist_cross_*: trees of touching slices exchanged, pair traversals): a JOINT call of all
        # ranks (collectives served in process), wall time of this rank's call
        # — the FIRST call (buffers allocated, code paged in) and, like the rebuild above, a second one
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cross = builder.cross_contacts(bvh)
        torch.cuda.synchronize()
        t_cross_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        cross = builder.cross_contacts(bvh)
        torch.cuda.synchronize()
        t_cross = time.perf_counter() - t0
        builder.time_phases = True
        builder.cross_contacts(bvh)
        builder.time_phases = False
        return {"rank": comm.rank, "leaves_in": n, "leaves_slice": last["n_slice"], "contacts": int(trav.num_contacts),
                "bytes_sent_to_peers": int(sum(c for r, c in enumerate(last["send_counts"]) if r != comm.rank) * rb),
                "bytes_received_from_peers": int(sum(c for r, c in enumerate(last["recv_counts"]) if r != comm.rank) * rb),
                "busiest_peer_bytes": int(max(c for r, c in enumerate(last["send_counts"]) if r != comm.rank) * rb),
                "joint_build_wall_ms": round(t_build * 1e3, 3), "traverse_alone_ms": round(t_trav * 1e3, 3),
                "cross_contacts": int(cross.shape[0]), "cross_partners": builder.last_cross["partners"],
                "cross_import_bytes": builder.last_cross["import_bytes"], "joint_cross_wall_ms": round(t_cross * 1e3, 3),
                "joint_cross_first_call_ms": round(t_cross_first * 1e3, 3), "joint_cross_phases_ms": builder.last_cross.get("phases_ms"),
                "splitter_key_bits": last["levels_used"]}
    rows = run_virtual_ranks(P, fn)
    slices = [r["leaves_slice"] for r in rows]
    line = {"metric": "BVH build+traverse throughput", "emulated": True, "n_gpus": 1, "virtual_ranks": P, "unit": "Mleaves/s",
            "value": None, "note": "EMULATED, unmeasured on xGMI: P virtual ranks (threads) share ONE GPU; per-rank figures show the shape of the work "
                                   "(exchange bytes per peer link, slice balance, traversal time of one slice alone on the GPU), not a scaling curve",
            "config": {"workload": f"BASELINE.json configs[4] law: {n * P} BSphere{{Float32}} leaves as {P} shards of {n}, distributed Morton + "
                                   "radix-sort exchange + global-AABB all-reduce (served in process), per-rank self-traverse"},
            "slice_imbalance": round(max(abs(s - n) for s in slices) / n, 5),
            "xgmi_link_time_at_153GBps_ms": round(max(r["busiest_peer_bytes"] for r in rows) / 153e9 * 1e3, 4),
            "per_rank": rows}
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: long enough for the GPU's clocks to settle — a 6 ms burst of 20 steps measures 4 % slower than the
    # steady state — and still a fraction of a second)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--n", type=int, default=0,
                    help="leaves per GPU (default: 1e6 at one GPU = BASELINE.json configs[1]; 12.5e6 at N > 1 = configs[4], "
                         "1e8 leaves over 8 GPUs)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE.json configs 2 (BFS), 3 and 4")
    ap.add_argument("--force-dist", action="store_true", help="use the multi-GPU build path even with one rank")
    ap.add_argument("--extra-n", type=int, default=10_000_000,
                    help="also report the north-star size (1e7 leaves) at N=1; 0 disables")
    ap.add_argument("--cpu-n", type=int, default=0, help="leaves of the CPU baseline sample (0 = same as --n, capped)")
    ap.add_argument("--detail", default="bench_detail.json", help="where the full (long) record goes; stdout carries the compact line")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="EMULATE BASELINE.json configs[4] on ONE GPU: this many virtual ranks (threads) x --n leaves (default 1.25e7) run the "
                         "distributed build + per-rank traversal; per-rank figures, no xGMI (nothing is measured about the links)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)  # (does not return)
    if args.virtual_ranks > 1:
        return emulate_config5(args)

    # the CPU baseline's OpenMP teams: one thread per core, neighbours first (read by libgomp when the oracle is loaded)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    import numpy as np
    import torch
    import implicitbvh_amd as ibvh
    from implicitbvh_amd import lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        import datetime
        # (a bounded timeout: a collective that cannot complete fails the run instead of hanging the node)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=300))
    lib.load()

    if world != max(args.gpus, 1) and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); running with {world}", file=sys.stderr)
    n = args.n or (1_000_000 if world == 1 else 12_500_000)
    n_global = n * world
    # SURVEY.md §8(d) config 2 law: r = r0*(0.5+0.5u), r0 = 0.5*(3k/(4 pi N))^(1/3), k = 8 -> ~1.8 contacts / leaf
    r0 = 0.5 * (3 * 8 / (4 * math.pi * n_global)) ** (1 / 3)
    vols = ibvh.generate_spheres(n, args.seed, first_index=rank * n, r0=r0)

    builder = None
    if dist is not None:
        from implicitbvh_amd import dist as ibvh_dist
        builder = ibvh_dist.DistributedBuilder(dist.group.WORLD)

        def one_step(state):
            bvh = builder.build(vols, cache=state[0])
            trav = ibvh.traverse(bvh, cache=state[1])
            return bvh, trav
    else:
        def one_step(state):
            bvh = ibvh.BVH(vols, cache=state[0])
            trav = ibvh.traverse(bvh, cache=state[1])
            return bvh, trav

    # ---- north-star size (1e7 leaves, single GPU): same step, fewer repetitions.  Runs BEFORE the headline's warm-up and timed
    # steps on purpose: a GPU that has just been idle runs a 5 ms burst ~12 % slower than its steady state (round 3: 0.2494 ms
    # in the driver's 20-step record against 0.223 ms over 200 steps), and these ~30 ms of work bring it to steady clocks; the
    # headline below is still W untimed warm-up steps followed by exactly K timed steps ---------------------
    north_star = None
    if world == 1 and dist is None and args.extra_n and args.extra_n != n:
        n2 = args.extra_n
        r02 = 0.5 * (3 * 8 / (4 * math.pi * n2)) ** (1 / 3)
        vols2 = ibvh.generate_spheres(n2, args.seed, r0=r02)
        st2 = (None, None)
        for _ in range(3):
            b2 = ibvh.BVH(vols2, cache=st2[0])
            st2 = (b2, ibvh.traverse(b2, cache=st2[1]))
            _ = st2[1].num_contacts
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            b2 = ibvh.BVH(vols2, cache=st2[0])
            st2 = (b2, ibvh.traverse(b2, cache=st2[1]))
            _ = st2[1].num_contacts
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        lib.call("ibvh_profile_enable", 1)
        for _ in range(3):
            b2 = ibvh.BVH(vols2, cache=st2[0])
            st2 = (b2, ibvh.traverse(b2, cache=st2[1]))
        torch.cuda.synchronize()
        prof2 = collect_profile(lib)
        lib.call("ibvh_profile_enable", 0)
        ms_phase = sum(prof2[k][0] for k in MORTON_SORT_KERNELS if k in prof2) / 3
        ms_build = sum(v[0] for k, v in prof2.items() if not k.startswith(TRAVERSE_PREFIXES)) / 3
        gb = 152.0 * n2 / (ms_phase * 1e-3) / 1e9
        dom2 = max(prof2, key=lambda k: prof2[k][0])
        avg2 = prof2[dom2][0] / prof2[dom2][1]
        ab2 = algorithmic_bytes(dom2, n2, st2[1].num_contacts)
        roof2 = _roof(dom2, avg2, ab2, "dominant kernel of the step at the north-star size (DESIGN.md §3 tables)") if ab2 else None
        if roof2 is not None:
            _with_block_kernel(roof2, {k: v[0] / 3 for k, v in prof2.items()}, ab2)
            roof2["cache_residency"] = "HBM: the step's working set (~1.3 GB) is 5x the 256 MB Infinity Cache"
            _attach_counters(roof2, "n1e7", avg2 * 1e-3)
        north_star = {"leaves": n2, "value": round(n2 * reps / el2 / 1e6, 3), "unit": "Mleaves/s", "roofline": roof2,
                      "ms_per_step": round(el2 / reps * 1e3, 4), "contacts": st2[1].num_contacts,
                      "build_ms": round(ms_build, 4),
                      "kernels_ms": {k: round(v[0] / 3, 4) for k, v in prof2.items()},
                      "morton_sort_phase": {"ms": round(ms_phase, 4), "algorithmic_GBps": round(gb, 1),
                                            "frac": round(gb / HBM_PEAK_GBS, 4), "bytes_per_leaf": 152}}
        del vols2, st2, b2
        torch.cuda.empty_cache()

    state = (None, None)
    for _ in range(args.warmup):
        state = one_step(state)
        _ = state[1].num_contacts

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- the timed region: EXACTLY K steps, each ending with the host's read of the contact count (what traverse() of the
    # reference returns, lvt/traverse_single.jl:60; SURVEY.md §8d times the step "incl. the count readback").  The read is a
    # poll of a pinned host word the scan kernel fills (include/ibvh.h, total_host): no stream sync, no device-to-host copy.
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = one_step(state)
        contacts = state[1].num_contacts
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    leaves_here = len(state[0].leaves)
    if dist is not None:
        c = torch.tensor([contacts, leaves_here], dtype=torch.int64, device="cuda")
        dist.all_reduce(c)
        contacts_total, leaves_total = int(c[0]), int(c[1])
    else:
        contacts_total, leaves_total = contacts, leaves_here
    ms_per_step = elapsed / args.steps * 1e3
    value = leaves_total * args.steps / elapsed / 1e6

    # The same K steps chained through cache= WITHOUT the read (a caller that sizes nothing from the count need not
    # wait for it; the count is read once, at the end)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = one_step(state)
    barrier()
    elapsed_eq = max_over_ranks(time.perf_counter() - t0)
    _ = state[1].num_contacts

    # ---- per-kernel timing pass (HIP events inside the library, on the launch stream) -------------
    prof_steps = max(3, min(10, args.steps))
    lib.call("ibvh_profile_enable", 1)
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    for _ in range(prof_steps):
        state = one_step(state)
    torch.cuda.synchronize()
    tp = time.perf_counter() - tp0
    prof = collect_profile(lib)
    lib.call("ibvh_profile_enable", 0)

    roofline, kernels = None, {}
    if prof:
        for k, (tot, num) in prof.items():
            avg_ms = tot / num
            ab = algorithmic_bytes(k, leaves_here, contacts)
            kernels[k] = {"avg_ms": round(avg_ms, 5), "launches_per_step": round(num / prof_steps, 2),
                          "ms_per_step": round(tot / prof_steps, 5),
                          "algorithmic_GBps": round(ab / (avg_ms * 1e-3) / 1e9, 1) if ab else None}
        dom = max(prof, key=lambda k: prof[k][0])
        avg_s = prof[dom][0] / prof[dom][1] * 1e-3
        ab = algorithmic_bytes(dom, leaves_here, contacts)
        achieved = ab / avg_s / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "algorithmic_bytes_per_launch": int(ab), "avg_launch_ms": round(avg_s * 1e3, 5)}
        _with_block_kernel(roofline, {k: v["ms_per_step"] for k, v in kernels.items()}, ab)
        # Morton+sort phase (north star: >= 40 % of the HBM roofline on 152 B/leaf)
        ms_phase = sum(v["ms_per_step"] for k, v in kernels.items() if k in MORTON_SORT_KERNELS)
        if ms_phase > 0:
            gbps = 152.0 * leaves_here / (ms_phase * 1e-3) / 1e9
            roofline["morton_sort_phase"] = {"ms": round(ms_phase, 4), "algorithmic_GBps": round(gbps, 1),
                                             "frac": round(gbps / HBM_PEAK_GBS, 4), "bytes_per_leaf": 152}

    # Counter figures of the dominant kernel (HBM-side traffic: 2*FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md §HBM
    # prescribes for gfx950; SQ instruction counters) come from separate rocprofv3 --pmc passes of this same command
    # (tools/profile_round.sh, tools/profile_sq.sh) committed under profiles/ and STAMPED with the hash of the kernel
    # sources they were taken at.  Measure-or-omit: a profile taken at another state of the kernels is not quoted.
    if roofline is not None and n in (1_000_000, 10_000_000):
        _attach_counters(roofline, "n1e6" if n == 1_000_000 else "n1e7", avg_s)
        if n == 1_000_000:
            roofline["cache_residency"] = ("infinity-cache-resident: the step's working set (~130 MB) fits the 256 MB Infinity Cache, so the "
                                           "'hbm' fraction at this size is a label, not a bound; north_star_1e7.roofline is the HBM-sized figure")

    # ---- per-rank exchange statistics of the distributed build (config 5): what the first real multi-GPU run needs to
    # price the all-to-all against xGMI (7 links x ~153 GB/s per GPU) -------------------------------------------------
    exchange = None
    if builder is not None:
        stats = builder.exchange_stats(vols, repeats=3)  # {"exchange_ms", "bytes_sent", "bytes_received", "peers"}
        rows = [None] * world
        dist.all_gather_object(rows, stats)
        if rank == 0:
            sent = [r["bytes_sent"] for r in rows]
            exchange = {"per_rank": rows, "max_exchange_ms": max(r["exchange_ms"] for r in rows),
                        "bytes_sent_max": max(sent),
                        "xgmi_frac_per_link": (round(max(sent) / max(world - 1, 1) / (max(r["exchange_ms"] for r in rows) * 1e-3) / 153e9, 4)
                                               if world > 1 else None),
                        "note": "all-to-all of 24-byte records, timed with HIP events around the collective on every rank; "
                                "xgmi_frac_per_link = bytes one GPU ships to ONE peer / time / 153 GB/s"}

    # ---- what makes the result GLOBAL: the contacts between leaves of different slices, which the
    # per-slice self-traversals of the timed step cannot see — timed on their own, every rank, max over
    # ranks; the phases of one step with a synchronisation between them; and, while the whole cloud fits one GPU, a self-check:
    # own + cross contacts of all ranks == the single-device contact count --------------------------------------------
    dist_detail = None
    if builder is not None:
        import torch.distributed as tdist
        cross = builder.cross_contacts(state[0])  # (untimed: sizes the buffers)
        barrier()
        reps_c = 3
        t0 = time.perf_counter()
        for _ in range(reps_c):
            cross = builder.cross_contacts(state[0])
        barrier()
        cross_s = max_over_ranks(time.perf_counter() - t0) / reps_c
        c = torch.tensor([int(cross.shape[0])], dtype=torch.int64, device="cuda")
        tdist.all_reduce(c)
        cross_total = int(c[0])
        builder.time_phases = True
        barrier()
        t0 = time.perf_counter()
        b_ph = builder.build(vols, cache=state[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr_ph = ibvh.traverse(b_ph, cache=state[1])
        _ = tr_ph.num_contacts
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        builder.cross_contacts(b_ph)  # (once more, synchronised between its four calls: last_cross["phases_ms"])
        builder.time_phases = False
        phases = dict(builder.last.get("phases_ms", {}))
        phases["traverse (per-slice LVT self-traverse + count read)"] = round((t2 - t1) * 1e3, 4)
        phases["cross-shard completion (plan + exchange + build + pair traverse, its own call)"] = round(cross_s * 1e3, 4)
        rows = [None] * world
        tdist.all_gather_object(rows, {"rank": rank, "phases_ms": phases, "cross": builder.last_cross})
        state = (b_ph, tr_ph)
        check = None
        if n_global <= 20_000_000:
            if rank == 0:
                single = ibvh.BVH(ibvh.generate_spheres(n_global, args.seed, r0=r0))
                want = int(ibvh.traverse(single).num_contacts)
                check = {"single_device_contacts": want, "own_plus_cross": contacts_total + cross_total,
                         "match": want == contacts_total + cross_total}
                del single
                torch.cuda.empty_cache()
            barrier()
        if rank == 0:
            keys = list(rows[0]["phases_ms"])
            dist_detail = {"cross_ms": round(cross_s * 1e3, 4), "contacts_cross_total": cross_total,
                           "contacts_total_with_cross": contacts_total + cross_total,
                           "phases_ms_max_over_ranks": {k: max(r["phases_ms"].get(k, 0.0) for r in rows) for k in keys},
                           "cross_bytes_received_max": max(r["cross"]["bytes_received"] for r in rows),
                           "per_rank": rows, "self_check": check,
                           "note": "the timed step (value) is distributed build + per-slice traversal, as BASELINE.json configs[4] words it; "
                                   "cross_ms is what completing the contact set costs on top"
                                   + ("" if world > 1 else "; ONE rank: no peers, nothing crosses")}

    # ---- CPU baseline: the oracle's multi-threaded restatement, rank 0 only, bounded sample --------
    cpu_baseline, cpu = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as orc  # the checker, here only as the timed CPU baseline
        ncpu = os.cpu_count() or 1
        cpu_n = args.cpu_n or min(n, 2_000_000)
        host = orc.generate_spheres_f32(cpu_n, args.seed, r0=0.5 * (3 * 8 / (4 * math.pi * cpu_n)) ** (1 / 3))
        # the baseline is the oracle's source built HERE with -O3 -march=native (strict IEEE flags kept,
        # so its contact count must equal the GPU's); the portable -O2 build the parity tests use is the fallback
        native = orc.load_native()
        # one thread first (the reference's CPU path with num_threads = 1), then a few team sizes: the host is shared and
        # SMT-threaded, more threads are not always faster; the best run is the baseline, its thread count is `cores`
        _, cc1, tb1, tt1 = orc.bench_build_traverse_f32(host, 1, native)
        phys = _host_cpu()["cores"] or ncpu  # one thread per PHYSICAL core at most: SMT siblings only add barrier cost here
        candidates = sorted({max(1, phys), max(1, phys // 2), max(1, phys // 4), min(32, phys)}, reverse=True)
        best, runs = None, []
        t_budget = time.perf_counter()
        for threads in candidates:
            for _ in range(3):
                _, cc, tb, tt = orc.bench_build_traverse_f32(host, threads, native)
                runs.append({"threads": threads, "mleaves_per_s": round(cpu_n / (tb + tt) / 1e6, 2)})
                if best is None or tb + tt < best[0] + best[1]:
                    best = (tb, tt, len(cc), threads)
            if time.perf_counter() - t_budget > 15:
                break
        cores = best[3]
        cpu = (orc, native, cores)
        host = _host_cpu()
        cpu_baseline = {"value": round(cpu_n / (best[0] + best[1]) / 1e6, 4), "unit": "Mleaves/s", "cores": cores,
                        "threads_used": cores, "host_cores": host["cores"], "host_threads": host["threads"], "cpu_model": host["model"],
                        "omp_binding": f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}",
                        "scaling_note": "`cores` is the thread count of the best run, not the host's core count (host_cores / host_threads): "
                                        "the sample is small (1e6 leaves: ~4,000 leaves per thread at 256 threads) and the stable LSB radix "
                                        "sort + the per-level merges are barrier-bound at that size, so large teams on this shared two-socket "
                                        "host lose to 32 - 64 threads; all runs are listed in noise.all_runs_mleaves_per_s by thread count",
                        "kind": "port",
                        "compiler_flags": "-O3 -march=native -ffp-contract=off (built on this host)" if native is not None
                                          else "-O2 -ffp-contract=off (portable build: the native build failed here)",
                        "sample": f"{cpu_n} BSphere{{Float32}} leaves, same generator/law as the GPU workload, "
                                  f"build {best[0]*1e3:.1f} ms + LVT traverse {best[1]*1e3:.1f} ms, {best[2]} contacts, "
                                  f"best run over thread counts {candidates} (<= 3 runs each; leaf ranges handed out "
                                  f"dynamically in the traversal), {cores} threads",
                        "noise": {"all_runs_mleaves_per_s": runs,
                                  "note": "shared 2-socket SMT host: run-to-run spread of the SAME code is up to 2x; the best run is "
                                          "quoted, so gpu_over_cpu is a LOWER bound on the typical ratio and good to +-2x at most"},
                        "build_ms": round(best[0] * 1e3, 3), "traverse_ms": round(best[1] * 1e3, 3),
                        "one_thread": {"value": round(cpu_n / (tb1 + tt1) / 1e6, 4), "build_ms": round(tb1 * 1e3, 2),
                                       "traverse_ms": round(tt1 * 1e3, 2), "contacts": len(cc1)},
                        "contacts_match_gpu": (best[2] == contacts and len(cc1) == contacts) if cpu_n == n else None,
                        "gpu_over_cpu": round(value / (cpu_n / (best[0] + best[1]) / 1e6), 1) if cpu_n == n else None,
                        "gpu_over_cpu_one_thread": round(value / (cpu_n / (tb1 + tt1) / 1e6), 1) if cpu_n == n else None}
        if north_star is not None:
            n2 = north_star["leaves"]
            host2 = orc.generate_spheres_f32(n2, args.seed, r0=0.5 * (3 * 8 / (4 * math.pi * n2)) ** (1 / 3))
            _, cc2, tb2, tt2 = orc.bench_build_traverse_f32(host2, cores, native)
            north_star["cpu_baseline"] = {"value": round(n2 / (tb2 + tt2) / 1e6, 4), "unit": "Mleaves/s", "cores": cores,
                                          "kind": "port", "build_ms": round(tb2 * 1e3, 2), "traverse_ms": round(tt2 * 1e3, 2),
                                          "sample": f"{n2} leaves, one run"}
            north_star["gpu_over_cpu"] = round(north_star["value"] / north_star["cpu_baseline"]["value"], 1)
            del host2

    # ---- work inflation of the headline traversal (SURVEY.md §8d "touched bytes"): node + leaf tests of the HIP walk (a
    # counting instantiation of the same kernel) next to the reference walk's (instrumented oracle), same input ----------
    work = None
    if cpu is not None and dist is None and n <= 2_000_000:
        orc, native, cores = cpu
        ob, _, _, _ = orc.bench_build_traverse_f32(orc.generate_spheres_f32(n, args.seed, r0=r0), cores, native)
        work = _work(ibvh, orc, ((state[0],), {}), ((ob,), {}), n, cores, native)
        del ob

    # ---- BASELINE.json configs 2 (BFS), 3, 4 ------------------------------------------------------------------------
    configs = None
    if rank == 0 and world == 1 and dist is None and not args.no_configs:
        del vols
        state = (None, None)
        torch.cuda.empty_cache()
        configs = run_configs(args, ibvh, lib, torch, cpu)

    if rank == 0:
        t_trav = sum(v["ms_per_step"] for k, v in kernels.items() if k.startswith(TRAVERSE_PREFIXES)) if kernels else None
        line = {
            "metric": "BVH build+traverse throughput", "value": round(value, 3), "unit": "Mleaves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n} random BSphere{{Float32}} leaves per GPU, BBox{{Float32}} nodes, UInt32 Morton, "
                                   f"Int32 index, build + LVT self-traverse + read of the contact count "
                                   + ("(BASELINE.json configs[1])" if world == 1 and n == 1_000_000 else
                                      f"(BASELINE.json configs[4] law: {n_global} leaves sharded over {world} GPU(s), distributed "
                                      f"Morton + radix-sort exchange, global-AABB RCCL all-reduce, per-GPU self-traverse)"
                                      if dist is not None else "(configs[1] law at another size)"),
                       "leaves_per_gpu": n, "leaves_total": leaves_total, "contacts_total": contacts_total,
                       "parallelism": "single GPU" if world == 1 else f"leaves sharded over {world} GPUs (RCCL build), per-GPU traversal"},
            "value_definition": "every timed step ends with the host's read of the contact count, as the reference's traverse() does "
                                "(lvt/traverse_single.jl:60); value_enqueue_only chains the same steps without it",
            "value_enqueue_only": round(leaves_total * args.steps / elapsed_eq / 1e6, 3),
            "ms_per_step_enqueue_only": round(elapsed_eq / args.steps * 1e3, 4),
            "mcontacts_per_s": round(contacts_total * args.steps / elapsed / 1e6, 3),
            "mcontacts_per_s_traverse_only": round(contacts / (t_trav * 1e-3) / 1e6, 3) if t_trav else None,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "work": work, "north_star_1e7": north_star, "configs": configs,
            "exchange": exchange, "dist": dist_detail, "kernels": kernels, "profiled_ms_per_step": round(tp / prof_steps * 1e3, 4),
        }
        # full detail beside the script (gitignored; tools/profile_round.sh copies it under profiles/), ONE compact line on stdout
        detail_path = args.detail
        try:
            with open(os.path.join(ROOT, detail_path) if not os.path.isabs(detail_path) else detail_path, "w") as f:
                json.dump(line, f, indent=1)
        except OSError as e:
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
            detail_path = None
        short = compact_line(line, detail_path)
        text = json.dumps(short, separators=(",", ":"))
        if len(text) > 4096:  # (never: the compact line has a fixed shape; if it ever grows, drop the per-config block first)
            short.pop("configs", None)
            text = json.dumps(short, separators=(",", ":"))
        print(text)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
