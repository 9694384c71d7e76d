import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lkgd_amd import ops, _lib
DEV="cuda:0"
def bench(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for kb in (32, 64, 128, 256):
    _lib.lib().lkgd_debug_set_gn_stats_kb(kb)
    line=[]
    for (H, W, C) in ((72, 128, 320), (72,128,960), (36, 64, 640), (18, 32, 1280), (9,16,1280)):
        T = 28 * H * W
        x = torch.randn(T, C, device=DEV, dtype=torch.float16)
        ms = min(bench(lambda: ops.groupnorm_stats(x, None, 28, H * W, 1e-5)) for _ in range(3))
        mt = min(bench(lambda: ops.groupnorm_stats(x, None, 2, 14 * H * W, 1e-5)) for _ in range(3))
        line.append(f"{H}x{W}x{C}: sp {ms*1e3:6.1f} us {T*C*2/ms/1e9:5.2f} TB/s | tm {mt*1e3:6.1f} us")
    print(f"stats chunk {kb:3d} KiB: " + " ; ".join(line), flush=True)
