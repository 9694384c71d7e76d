import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import replay_probe as rp
from lkgd_amd import replay, ops
import lkgd_amd.pipeline as pl

orig = replay.record
class rec_arena_keepall(orig):
    def __init__(self, arena=None, keep_all=None):
        super().__init__(arena, True)

def cold():
    ops._zeros.clear()
    ops._tls.splitk_ws = {}

for mode in ("arena_keepall", "arena_free", "arena_free_warm_zeros", "arena_free_warm_ws", "arena_free_warm_both"):
    cold()
    if "warm_zeros" in mode or "warm_both" in mode:
        ops.zeros_page(torch.device("cuda", 0))
    if "warm_ws" in mode or "warm_both" in mode:
        ops.splitk_workspace(torch.device("cuda", 0))
    pipe, run = rp.variant("stock")
    pl._replay.record = rec_arena_keepall if mode == "arena_keepall" else orig
    pipe.use_replay = True
    a = run().float().cpu(); a2 = run().float().cpu()
    pipe.use_replay = False
    b = run().float().cpu()
    print(f"{mode:28s} a==b {torch.equal(a, b)} a2==b {torch.equal(a2, b)} a==a2 {torch.equal(a, a2)} "
          f"|a-b| {(a - b).abs().max().item():.3e} nan {torch.isnan(a).any().item()}", flush=True)
