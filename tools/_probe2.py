import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import replay_probe as rp
from lkgd_amd import replay
import lkgd_amd.pipeline as pl

pipe, run = rp.variant("stock")
pipe.use_replay = False
e1 = run().float().cpu(); e2 = run().float().cpu()
print("E1 eager==eager", torch.equal(e1, e2), flush=True)
orig = replay.record
class rec_keepall_noarena(orig):
    def __init__(self, arena=None, keep_all=None):
        if arena is not None: arena.busy = False
        super().__init__(None, True)
class rec_arena_keepall(orig):
    def __init__(self, arena=None, keep_all=None):
        super().__init__(arena, True)
pipe.use_replay = True
pl._replay.record = rec_keepall_noarena
r = run().float().cpu(); print("E2 replay(keep-all, no arena)==eager", torch.equal(r, e1), flush=True)
pl._replay.record = rec_arena_keepall
r = run().float().cpu(); print("E4 replay(arena, keep-all)==eager", torch.equal(r, e1), flush=True)
pl._replay.record = orig
r = run().float().cpu(); print("E3 replay(arena, free scratch)==eager", torch.equal(r, e1), flush=True)
r2 = run().float().cpu(); print("E3b again", torch.equal(r2, e1), torch.equal(r, r2), flush=True)
# pool isolation
a = pipe._arenas._arenas[0]
pid = tuple(a.pool.id)
snap = torch.cuda.memory_snapshot()
segs = [(s["address"], s["address"] + s["total_size"]) for s in snap if tuple(s.get("segment_pool_id", (0, 0))) == pid]
print("pool id", pid, "segments", len(segs), "use_count", a.pool.use_count(), flush=True)
ts = [torch.empty(n, device="cuda") for n in (64, 1 << 12, 1 << 18, 1 << 22, 1 << 24) for _ in range(20)]
bad = sum(1 for t in ts for lo, hi in segs if lo <= t.data_ptr() < hi)
print("general allocations inside pool segments:", bad, flush=True)
