"""In-process virtual ranks: `size` Python threads of ONE process, each a rank with tensor-level collectives (all_reduce,
all_gather, all_to_all) between them.  Test / emulation scaffolding only (tests/, bench.py --virtual-ranks): a single GPU
then plays all the GPUs of a node — same kernels, same order of collectives, no xGMI."""
import threading


class ThreadWorld:
    def __init__(self, size):
        self.size = size
        self.barrier = threading.Barrier(size)
        self.slots = [None] * size


class ThreadComm:
    """Collectives between `size` Python threads of one process (one virtual rank each)."""

    def __init__(self, world, rank):
        self.world, self.rank, self.size = world, rank, world.size

    def all_reduce(self, t, op):
        import torch
        w = self.world
        w.slots[self.rank] = t.clone()
        w.barrier.wait()
        stack = torch.stack(w.slots)
        res = {"min": lambda s: s.min(0).values, "max": lambda s: s.max(0).values, "sum": lambda s: s.sum(0)}[op](stack)
        w.barrier.wait()
        t.copy_(res)
        return t

    def all_gather(self, t):
        import torch
        w = self.world
        w.slots[self.rank] = t.clone()
        w.barrier.wait()
        out = torch.stack(w.slots)
        w.barrier.wait()
        return out

    def all_to_all(self, send, send_counts, recv_counts):
        import torch
        w = self.world
        w.slots[self.rank] = (send, [int(c) for c in send_counts])
        w.barrier.wait()
        pieces = []
        for src in range(self.size):
            s, sc = w.slots[src]
            off = sum(sc[:self.rank])
            pieces.append(s[off:off + sc[self.rank]])
            assert sc[self.rank] == int(recv_counts[src])
        out = torch.cat(pieces) if pieces else send[:0]
        w.barrier.wait()
        return out


def run_virtual_ranks(size, fn):
    """Run fn(comm) on `size` virtual ranks (threads); returns the list of results in rank order."""
    world = ThreadWorld(size)
    results, errors = [None] * size, []

    def work(r):
        try:
            results[r] = fn(ThreadComm(world, r))
        except BaseException as e:  # noqa: BLE001 - re-raised below
            errors.append(e)
            world.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results
