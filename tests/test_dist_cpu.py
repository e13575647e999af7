"""Multi-rank build logic on CPU: world_size-2 gloo processes (and in-process virtual ranks) drive the CPU stand-in of the
distributed build's driver (tests/dist_cpu_driver.py: the same collectives, the PRODUCT's host-only splitter arithmetic
ibvh_splitter_search_* of libibvh, an oracle-backed engine standing in for the HIP kernels), so the splitter search, send
matrix, exchange and global numbering are exercised without a GPU; the device driver itself (csrc/ibvh_distdrv.hip) is
covered on the GPU with virtual ranks and with RCCL at world size 1 (tests/test_gpu_parity.py, tests/test_gpu_dist.py).  Property:
concatenating the ranks' sorted leaves reproduces the single-device sorted leaf array bit for bit."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as orc
import implicitbvh_amd as ibvh
from implicitbvh_amd import abi
from implicitbvh_amd import dist as ibd
from dist_cpu_driver import CpuDistributedBuilder
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from virtual_ranks import run_virtual_ranks  # noqa: E402


class OracleEngine:
    """CPU stand-in for HipEngine built on the oracle (test infrastructure only)."""
    device = "cpu"

    def tensor(self, data, dtype):
        return torch.tensor(data, dtype=dtype)

    def _vols(self, types, vols):
        return orc.as_volumes(vols.numpy(), types.leaf_kind, types.leaf_float)

    def extrema(self, types, vols):
        return torch.from_numpy(orc.extrema(types, self._vols(types, vols), wrapped=False, expand=False))

    def expand(self, types, ext):
        f = abi.FLOAT_DTYPES[types.leaf_float]
        rp = f(1e-5) if f is np.float32 else f(1e-14)
        fm = np.finfo(f).tiny
        e = ext.numpy()
        for k in range(6):
            a = f(rp * np.abs(e[k]))
            e[k] = f(f(e[k] - a) - fm) if k < 3 else f(f(e[k] + a) + fm)
        return ext

    def keys(self, types, vols, ext):
        k = orc.morton_keys(types, self._vols(types, vols), False, ext.numpy())
        return torch.from_numpy(k.astype(np.int64 if k.dtype == np.uint64 else np.int32))

    def histogram(self, keys, shift, bits, prefix_shift, prefixes):
        k = keys.numpy().astype(np.uint64)
        if len(k) == 0:
            return torch.zeros((max(len(prefixes), 1), 1 << bits), dtype=torch.int64)
        d = ((k >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
        if not prefixes:
            return torch.from_numpy(np.bincount(d, minlength=1 << bits)[None, :].astype(np.int64))
        p = k >> np.uint64(prefix_shift)
        rows = [np.bincount(d[p == np.uint64(v)], minlength=1 << bits) for v in prefixes]
        return torch.from_numpy(np.stack(rows).astype(np.int64))

    def partition(self, keys, splitters, nranks, known_counts=None):
        k = keys.numpy()
        dest = np.searchsorted(np.asarray(splitters, dtype=k.dtype), k, side="right") if splitters else np.zeros(len(k), int)
        perm = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=nranks).tolist()
        assert known_counts is None or list(known_counts) == counts  # the send matrix derived from histograms
        return torch.from_numpy(perm.astype(np.int32)), counts

    def pack(self, types, vols, keys, perm, index_base):
        dt = abi.leaf_dtype(types)
        rec = np.zeros(vols.shape[0], dt)
        p = perm.numpy().astype(np.int64)
        rec["volume"] = self._vols(types, vols)[p]
        rec["index"] = index_base + p + 1
        rec["morton"] = keys.numpy()[p].astype(dt["morton"])
        return torch.from_numpy(rec.view(np.uint8).copy()), dt.itemsize

    def build_local(self, types, records, n, ext_host, node_type, options, cache):
        rec = records.numpy().view(abi.leaf_dtype(types))
        v = rec["volume"]
        vols = np.concatenate([v[f].reshape(n, -1) for f in v.dtype.names], axis=1)
        return orc.build(vols, types, indices=rec["index"], compute_extrema=False, mins=ext_host[:3], maxs=ext_host[3:])

    def to_host(self, t):
        return t.numpy()

    # cross-shard completion
    def root_box(self, bvh):
        if len(bvh.nodes):
            v = bvh.nodes[0]
            return torch.tensor(np.concatenate([v["lo"], v["up"]]).astype(np.float64))
        s = bvh.leaves["volume"][0]
        return torch.tensor(np.concatenate([s["x"] - s["r"], s["x"] + s["r"]]).astype(np.float64))

    def export(self, bvh):
        return torch.from_numpy(np.concatenate([bvh.leaves.view(np.uint8), bvh.nodes.view(np.uint8)]).copy())

    def export_bytes(self, types, n):
        t = orc.tree_shape(n)
        return n * abi.leaf_dtype(types).itemsize + (t.real_nodes - t.real_leaves) * abi.node_dtype(types).itemsize

    def import_(self, types, n, buf):
        t = orc.tree_shape(n)
        lb = n * abi.leaf_dtype(types).itemsize
        leaves = buf.numpy()[:lb].copy().view(abi.leaf_dtype(types))
        nodes = buf.numpy()[lb:].copy().view(abi.node_dtype(types))
        skips = orc.compute_skips(t).astype(abi.INDEX_DTYPES[types.index_type])
        return orc.HostBVH(types, t, 1, leaves, nodes, skips, None)

    def pair_contacts(self, a, b):
        c = orc.traverse_pair_lvt(a, b)[0]
        return torch.from_numpy(np.stack([c["a"], c["b"]], axis=1).astype(np.int64).reshape(-1, 2))

    def empty_contacts(self, types):
        return torch.zeros((0, 2), dtype=torch.int64)

    def cat(self, ts):
        return torch.cat(ts)


def cloud(n, seed, kind=abi.BSPHERE, dtype=np.float32):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3)) ** 2 * 10 - 3  # skewed, partly negative
    if kind == abi.BSPHERE:
        return np.concatenate([c, 0.01 + 0.1 * rng.random((n, 1))], axis=1).astype(dtype)
    h = 0.01 + 0.1 * rng.random((n, 3))
    return np.concatenate([c - h, c + h], axis=1).astype(dtype)


def shard_bounds(n, world):
    return [n * r // world for r in range(world + 1)]


def check_against_single_build(vols, types, per_rank_leaves):
    single = orc.build(vols, types)
    cat = np.concatenate(per_rank_leaves)
    assert len(cat) == len(vols)
    assert cat["morton"].tolist() == single.leaves["morton"].tolist()
    assert cat["index"].tolist() == single.leaves["index"].tolist()  # incl. stable tie order
    assert cat["volume"].tobytes() == single.leaves["volume"].tobytes()


def _worker(rank, world, path, n, seed, kind, flt, morton, init_file):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        dtype = abi.FLOAT_DTYPES[flt]
        vols = cloud(n, seed, kind, dtype)
        b = shard_bounds(n, world)
        local = torch.from_numpy(vols[b[rank]:b[rank + 1]].copy())
        types = abi.make_types(kind, flt, abi.BBOX, abi.F32, abi.I32, morton)
        opts = ibvh.BVHOptions(morton=ibvh.DefaultMortonAlgorithm(abi.MORTON_DTYPES[morton]))
        builder = CpuDistributedBuilder(ibd.TorchComm(), OracleEngine())
        bvh = builder.build(local, ibvh.api._VolumeType(abi.BBOX, abi.F32), options=opts)
        np.save(os.path.join(path, f"leaves_{rank}.npy"), bvh.leaves)
        np.save(os.path.join(path, f"ext_{rank}.npy"), builder.last["extrema"])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,flt,morton,n", [(abi.BSPHERE, abi.F32, abi.U32, 20011), (abi.BBOX, abi.F64, abi.U64, 5003),
                                               (abi.BSPHERE, abi.F32, abi.U16, 3001)])
def test_gloo_world2_matches_single_device_build(kind, flt, morton, n):
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker, args=(world, tmp, n, 1234, kind, flt, morton, init_file), nprocs=world, join=True)
        leaves = [np.load(os.path.join(tmp, f"leaves_{r}.npy")) for r in range(world)]
        exts = [np.load(os.path.join(tmp, f"ext_{r}.npy")) for r in range(world)]
    types = abi.make_types(kind, flt, abi.BBOX, abi.F32, abi.I32, morton)
    vols = cloud(n, 1234, kind, abi.FLOAT_DTYPES[flt])
    assert exts[0].tobytes() == exts[1].tobytes() == orc.build(vols, types).extrema.tobytes()
    check_against_single_build(vols, types, leaves)
    assert abs(len(leaves[0]) - len(leaves[1])) <= max(8, n // 100)  # balanced within the splitter tolerance


def ragged_clustered_cloud(n, seed):
    """Three tight Gaussian clusters + a thin uniform background + exact duplicates: crowded Morton cells, splitters
    that need refinement, ties across ranks."""
    rng = np.random.default_rng(seed)
    centres = rng.random((3, 3)) * 8 - 2
    which = rng.integers(0, 3, n)
    c = centres[which] + rng.normal(0, 0.01, (n, 3))
    bg = rng.random(n) < 0.1
    c[bg] = rng.random((int(bg.sum()), 3)) * 10 - 3
    c[n // 2: n // 2 + 200] = c[0]  # duplicates of one leaf, spread over ranks by the ragged sharding below
    return np.concatenate([c, 0.005 + 0.01 * rng.random((n, 1))], axis=1).astype(np.float32)


def _worker_ragged(rank, world, path, n, seed, bounds, init_file):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        vols = ragged_clustered_cloud(n, seed)
        local = torch.from_numpy(vols[bounds[rank]:bounds[rank + 1]].copy())
        builder = CpuDistributedBuilder(ibd.TorchComm(), OracleEngine())
        bvh = builder.build(local)
        np.save(os.path.join(path, f"leaves_{rank}.npy"), bvh.leaves)
        np.save(os.path.join(path, f"send_{rank}.npy"), np.asarray(builder.last["send_counts"]))
        np.save(os.path.join(path, f"recv_{rank}.npy"), np.asarray(builder.last["recv_counts"]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gloo_world4_ragged_clustered_shards():
    """Four real processes over gloo, shards of very different sizes (one of them EMPTY: it still takes part in every
    collective), clustered keys with duplicates: concatenated slices == the single-device build, and the exchange sizes
    every rank derived are consistent (send[r][d] == recv[d][r])."""
    world, n, seed = 4, 24_019, 77
    bounds = [0, 0, 301, 16_000, n]  # rank 0 holds nothing, rank 1 a handful, rank 2 most
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker_ragged, args=(world, tmp, n, seed, bounds, init_file), nprocs=world, join=True)
        leaves = [np.load(os.path.join(tmp, f"leaves_{r}.npy")) for r in range(world)]
        send = np.stack([np.load(os.path.join(tmp, f"send_{r}.npy")) for r in range(world)])
        recv = np.stack([np.load(os.path.join(tmp, f"recv_{r}.npy")) for r in range(world)])
    vols = ragged_clustered_cloud(n, seed)
    check_against_single_build(vols, abi.make_types(), leaves)
    assert np.array_equal(send, recv.T)
    assert send[0].sum() == 0 and send.sum() == n
    sizes = [len(x) for x in leaves]
    assert min(sizes) >= 1 and max(sizes) - min(sizes) <= max(8, n // 50)


@pytest.mark.parametrize("world", [1, 3, 8])
def test_virtual_ranks_cpu(world):
    n = 30011
    vols = cloud(n, 99)
    types = abi.make_types()
    b = shard_bounds(n, world)

    def fn(comm):
        builder = CpuDistributedBuilder(comm, OracleEngine())
        bvh = builder.build(torch.from_numpy(vols[b[comm.rank]:b[comm.rank + 1]].copy()))
        return bvh.leaves, builder.last
    out = run_virtual_ranks(world, fn)
    check_against_single_build(vols, types, [o[0] for o in out])
    sizes = [len(o[0]) for o in out]
    assert max(sizes) - min(sizes) <= max(8, n // (50 * world))
    assert all(o[1]["splitters"] == out[0][1]["splitters"] for o in out)


@pytest.mark.parametrize("world", [2, 4])
def test_cross_shard_completion_cpu(world):
    """Per-slice self contacts + cross-slice pair contacts == contact set of the single-device build."""
    n = 6007
    rng = np.random.default_rng(21)
    vols = np.concatenate([rng.random((n, 3)) * 4, 0.05 + 0.1 * rng.random((n, 1))], axis=1).astype(np.float32)
    types = abi.make_types()
    single = orc.build(vols, types)
    c = orc.traverse_lvt(single)[0]
    want = set(zip(c["a"].tolist(), c["b"].tolist()))
    b = shard_bounds(n, world)

    def fn(comm):
        builder = CpuDistributedBuilder(comm, OracleEngine())
        bvh = builder.build(torch.from_numpy(vols[b[comm.rank]:b[comm.rank + 1]].copy()))
        own = orc.traverse_lvt(bvh)[0]
        cross = builder.cross_contacts(bvh).numpy()
        return set(zip(own["a"].tolist(), own["b"].tolist())), {(min(x, y), max(x, y)) for x, y in cross.tolist()}
    out = run_virtual_ranks(world, fn)
    got = set()
    total = 0
    for own, cross in out:
        total += len(own) + len(cross)
        got |= own | cross
    assert total == len(got) and got == want and len(want) > n


def test_exact_splitters_with_zero_tolerance():
    n, world = 20011, 4
    vols = cloud(n, 3)
    b = shard_bounds(n, world)

    def fn(comm):
        builder = CpuDistributedBuilder(comm, OracleEngine(), tolerance=0.0)
        return builder.build(torch.from_numpy(vols[b[comm.rank]:b[comm.rank + 1]].copy())).leaves
    out = run_virtual_ranks(world, fn)
    check_against_single_build(vols, abi.make_types(), out)
    sizes = [len(o) for o in out]
    assert max(sizes) - min(sizes) <= 4  # exact up to the multiplicity of one key


def test_heavy_duplicates_and_uneven_shards():
    """Few distinct keys (ties must keep global input order) and ranks with very different shard sizes."""
    rng = np.random.default_rng(5)
    base = cloud(23, 7)
    vols = np.repeat(base, 400, axis=0)
    rng.shuffle(vols)
    n = len(vols)
    bounds = [0, 17, 5000, n]

    def fn(comm):
        builder = CpuDistributedBuilder(comm, OracleEngine())
        return builder.build(torch.from_numpy(vols[bounds[comm.rank]:bounds[comm.rank + 1]].copy())).leaves
    out = run_virtual_ranks(3, fn)
    check_against_single_build(vols, abi.make_types(), out)


def test_starved_rank_raises_on_every_rank():
    """All leaves share one Morton key: every splitter coincides, one rank would receive everything and the others
    nothing.  Every rank must raise (a rank that carried on alone would hang in the next collective)."""
    one = cloud(1, 3)
    vols = np.repeat(one, 600, axis=0)
    seen = []

    def fn(comm):
        builder = CpuDistributedBuilder(comm, OracleEngine())
        try:
            builder.build(torch.from_numpy(vols[comm.rank * 200:(comm.rank + 1) * 200].copy()))
        except abi.DomainError:
            seen.append(comm.rank)
            return "raised"
        return "built"
    out = run_virtual_ranks(3, fn)
    assert out == ["raised"] * 3 and sorted(seen) == [0, 1, 2]
