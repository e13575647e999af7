"""Debug probe of the finish kernel's rescue path: cold build (two levels) vs the same input rebuilt with sort_levels = 0."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import implicitbvh_amd as ibvh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
rng = np.random.default_rng(n)
c = 0.5 + 1e-3 * rng.normal(0, 1, (n, 3)); c[0] = 100.0
v = torch.from_numpy(np.concatenate([c, 1e-4 * rng.random((n, 1))], axis=1).astype(np.float32)).cuda()
g = ibvh.BVH(v)
torch.cuda.synchronize()
ref_idx = g.leaves.index.clone(); ref_m = g.leaves.morton.clone()
print("cold hint", hex(int(ibvh.api._host_words().words[g._skew.slot])))
for trial in range(1):
    g._skew[0] = 0
    g = ibvh.BVH(v, cache=g)
    torch.cuda.synchronize()
    idx = g.leaves.index; m = g.leaves.morton
    bad = (idx != ref_idx).nonzero().flatten().cpu().numpy()
    print("trial", trial, "morton equal", bool((m == ref_m).all()), "sorted", bool((m[1:] >= m[:-1]).all()), "index mismatches", len(bad))
    sc = g._scratch.view(torch.int32).cpu().numpy().view(np.uint32)
    at = np.nonzero(sc == 0xDEADBEEF)[0]
    print("magic at", at, "scratch words", len(sc))
    seq = np.nonzero((sc[:-2] == 4096 + 5) & (sc[1:-1] == 4096 + 6) & (sc[2:] == 4096 + 7))[0]
    print("gather-list-like sequences at", seq[:10])
    if len(seq):
        b0 = seq[0] - 5
        print(" list head", sc[b0:b0 + 8], "entries 168..172", sc[b0 + 168:b0 + 173], "g_first head", sc[b0 + 2048:b0 + 2056])
    if len(at):
        d = sc[at[-1]:at[-1] + 1024]
        print(" entry t seen by thread t:", d[1:12], d[60:70], d[165:175], d[250:257])
        print(" entry 0 seen by thread t:", d[300:312], d[364:368], d[550:556])
        print(" mine of thread t:", d[600:612], d[664:668], d[850:856])
        print(" chunks seen by thread 5:", d[900:916], "by thread 0:", d[920:936])
    if len(bad):
        print(" first", bad[:10], "last", bad[-5:])
        print(" got", idx[bad[:12]].cpu().numpy(), "want", ref_idx[bad[:12]].cpu().numpy())
        d = np.diff(bad); brk = np.nonzero(d > 1)[0]
        print(" runs of mismatches start at", bad[np.r_[0, brk + 1]][:20], "count", len(brk) + 1)
        u = torch.unique(idx).numel()
        print(" unique indices", u, "of", n)
        km = ref_m.cpu().numpy(); first = int(np.argmax(np.bincount((km >> 19))[:]))  # rough: the crowded 11-bit cell
        cell = km >> 19; starts = np.nonzero(np.r_[True, cell[1:] != cell[:-1]])[0]; sizes = np.diff(np.r_[starts, n])
        big = starts[np.argmax(sizes)]; print(" crowded cell starts at", big, "size", sizes.max(), "distinct keys in it", len(np.unique(km[big:big + sizes.max()])))
