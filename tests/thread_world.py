"""A stand-in for ``torch.distributed`` that runs every rank of a world as a THREAD of one process (test infrastructure).

Why: a GPU box of this pool admits six processes on its card, pytest included, so the eight-rank CFG x (4,4,3,3) layout of
DESIGN.md section 6 cannot run as eight processes there.  The sharded denoiser talks to ``torch.distributed`` through a handful
of calls (lkgd_amd/dist.py, dist_run.py: new_group, all_gather_into_tensor, all_to_all_single, all_reduce, get_backend,
is_initialized); this module implements exactly those over ``threading.Barrier`` and device-to-device copies.  All threads
launch on the same (default) stream, so a peer's tensor deposited before a barrier is complete - in stream order - by the time
another thread's copy of it is enqueued after the barrier.  Reductions add in group-rank order (deterministic)."""
import threading


class _Group:
    def __init__(self, ranks):
        self.ranks = list(ranks)
        self.barrier = threading.Barrier(len(self.ranks))
        self.slots = [None] * len(self.ranks)


class ReduceOp:
    SUM = "sum"


class ThreadWorld:
    ReduceOp = ReduceOp

    def __init__(self, world: int):
        self.world = world
        self._tls = threading.local()
        self._lock = threading.Lock()
        self._groups = {}
        self._world_group = _Group(range(world))

    # ---- per-thread identity ------------------------------------------------------------------------------------------
    def bind(self, rank: int) -> None:
        self._tls.rank = rank

    def get_rank(self, group=None) -> int:
        g = group or self._world_group
        return g.ranks.index(self._tls.rank)

    def get_world_size(self, group=None) -> int:
        return len((group or self._world_group).ranks)

    def is_initialized(self) -> bool:
        return True

    def get_backend(self, group=None) -> str:
        return "nccl"                       # device tensors go through as they are (no host staging)

    def new_group(self, ranks):
        """every thread calls this with the same lists in the same order; the object is shared"""
        key = tuple(ranks)
        with self._lock:
            if key not in self._groups:
                self._groups[key] = _Group(ranks)
            return self._groups[key]

    def barrier(self, group=None, **_):
        (group or self._world_group).barrier.wait()

    def destroy_process_group(self):
        pass

    # ---- collectives -----------------------------------------------------------------------------------------------------
    def _enter(self, group, item):
        g = group or self._world_group
        i = g.ranks.index(self._tls.rank)
        g.slots[i] = item
        g.barrier.wait()
        return g, i

    def all_gather_into_tensor(self, out, inp, group=None):
        g, _ = self._enter(group, inp)
        n = inp.shape[0]
        for p, t in enumerate(g.slots):
            out[p * n:(p + 1) * n].copy_(t)
        g.barrier.wait()

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        g, i = self._enter(group, (inp, list(input_split_sizes)))
        o = 0
        for p, (pinp, prow) in enumerate(g.slots):
            start, n = sum(prow[:i]), prow[i]
            assert n == output_split_sizes[p], (n, output_split_sizes, p)
            out[o:o + n].copy_(pinp[start:start + n])
            o += n
        g.barrier.wait()
        if async_op:            # the issue / wait split of lkgd_amd.dist (asynchronous with RCCL): here the copies are enqueued on the
            return _Done()      # one shared stream at the issue point, so the "work" has nothing left to wait for

    def all_reduce(self, t, op=ReduceOp.SUM, group=None):
        assert op == ReduceOp.SUM
        g, _ = self._enter(group, t.clone())
        acc = g.slots[0].clone()
        for s in g.slots[1:]:
            acc += s
        g.barrier.wait()
        t.copy_(acc)


class _Done:
    """what torch.distributed returns for async_op=True, as far as lkgd_amd.dist uses it"""

    def wait(self):
        return True


def run_ranks(tw: ThreadWorld, fn):
    """fn(rank) on tw.world threads; returns the list of results in rank order, re-raises the first exception"""
    world = tw.world
    out, err = [None] * world, []

    def body(r):
        try:
            tw.bind(r)
            out[r] = fn(r)
        except BaseException as e:          # noqa: BLE001 - reported to the caller; the other threads must not hang forever
            err.append((r, e))
            for g in [tw._world_group] + list(tw._groups.values()):
                g.barrier.abort()

    ths = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if err:
        real = [e for e in err if not isinstance(e[1], threading.BrokenBarrierError)] or err
        raise RuntimeError(f"rank {real[0][0]} failed: {real[0][1]!r}") from real[0][1]
    return out
