"""The reference's literal time-stepping loop (build.jl:109-126, README.md:84-95) through the HIP library:

    bvh = BVH(wrapped_leaves, N)                 # user indices
    loop:  move bvh.leaves[i].volume IN PLACE;  bvh = BVH(bvh.leaves, N; cache=bvh);  traversal = traverse(bvh; cache=traversal)

The record array is input AND output of every rebuild, the user indices travel with the records, the input of step k is
the Morton order of step k-1 displaced by at most a cell (nearly sorted).  Every step is compared with the oracle fed the
SAME chain: sorted records (volume, index, Morton code), every node, and the LVT contact list including its order."""
import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import implicitbvh_amd as ibvh  # noqa: E402
from implicitbvh_amd import abi  # noqa: E402


def _chain(n, steps, cells, seed, leaf_kind=abi.BSPHERE):
    rng = np.random.default_rng(seed)
    types = abi.make_types(leaf_kind, abi.F32, abi.BBOX, abi.F32)
    r0 = 0.5 * (3 * 8 / (4 * np.pi * n)) ** (1 / 3)
    c = rng.random((n, 3)).astype(np.float32)
    if leaf_kind == abi.BSPHERE:
        vols = np.concatenate([c, (r0 * (0.5 + 0.5 * rng.random((n, 1)))).astype(np.float32)], axis=1)
    else:
        h = (r0 * (0.5 + 0.5 * rng.random((n, 3)))).astype(np.float32)
        vols = np.concatenate([c - h, c + h], axis=1)
    user = rng.permutation(n).astype(np.int32) * 3 + 7  # arbitrary distinct user indices: they must survive every rebuild
    bv = ibvh.BoundingVolumes.wrap(torch.from_numpy(vols).cuda(), user)
    g = ibvh.BVH(bv)
    o = orc.build(vols, types, indices=user)
    trav = None
    step = np.float32(cells / 1024.0)
    for k in range(steps + 1):
        gl = g.leaves.to_numpy()
        assert gl["index"].tolist() == o.leaves["index"].tolist(), f"step {k}: user indices / order differ"
        assert gl["morton"].tolist() == o.leaves["morton"].tolist()
        assert gl["volume"].tobytes() == o.leaves["volume"].tobytes()
        gn = g.nodes.cpu().numpy()
        assert gn.tobytes() == o.nodes.view(gn.dtype).reshape(gn.shape).tobytes(), f"step {k}: nodes differ"
        trav = ibvh.traverse(g, cache=trav)
        exp, _ = orc.traverse_lvt(o)
        got = trav.contacts.cpu().numpy()
        assert got.shape[0] == len(exp) and (got[:, 0] == exp["a"]).all() and (got[:, 1] == exp["b"]).all(), f"step {k}: contact list differs"
        if k == steps:
            break
        # move every leaf in place (the same float32 arithmetic on both sides), then rebuild from the moved, Morton-ordered records
        w = 3 if leaf_kind == abi.BSPHERE else 6
        delta = ((rng.random((n, 3)) * 2 - 1).astype(np.float32) * step)
        dfull = np.zeros((n, w), np.float32)
        dfull[:, :3] = delta
        if leaf_kind == abi.BBOX:
            dfull[:, 3:] = delta
        before = g.leaves.buf.data_ptr()
        g.leaves.volume[:, :w] += torch.from_numpy(dfull).cuda()
        hv = np.ascontiguousarray(gl["volume"]).view(np.float32).reshape(n, -1).copy()
        hv[:, :w] += dfull
        g = ibvh.BVH(g.leaves, cache=g)  # in place: same record array in and out
        assert g.leaves.buf.data_ptr() == before
        o = orc.build(hv, types, indices=gl["index"])
    return trav


@pytest.mark.parametrize("n", [1000, 20000, 300000])
def test_in_place_time_stepping_matches_the_oracle_chain(n):
    trav = _chain(n, steps=5, cells=1.0, seed=n)
    assert trav.num_contacts > 0


def test_in_place_time_stepping_boxes_and_big_moves():
    """BBox leaves, and moves of 40 cells per step (the input is then far from sorted): same bar."""
    _chain(50000, steps=3, cells=40.0, seed=5, leaf_kind=abi.BBOX)
    _chain(50000, steps=3, cells=40.0, seed=6)
