import sys, os
sys.path.insert(0, os.getcwd())
import torch
from lkgd_amd import _lib, ops
L = _lib.lib()
DEV = "cuda:0"
g = torch.Generator().manual_seed(1)
def run(M, N, K, res=True, bias=True):
    a = (torch.randn(M, K, generator=g)).half(); w = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g).half()
    ref = a.float() @ w.float().T + (b if bias else 0) + (r.float() if res else 0)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
    L.lkgd_debug_set_gemm_variant(6)
    ops.gemm(a.to(DEV), w.to(DEV), out, M=M, N=N, K=K, bias=b.to(DEV) if bias else None, res1=r.to(DEV) if res else None)
    torch.cuda.synchronize()
    d = (out.float().cpu() - ref).abs()
    bad = d > 0.05
    print(f"M={M} N={N} K={K} res={res} bias={bias}: max err {d.max().item():.3f} nan {torch.isnan(out).sum().item()} bad {bad.sum().item()} / {bad.numel()}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("   bad rows", rows[:20].tolist(), "... n", len(rows), " rows%32:", sorted(set((rows % 32).tolist()))[:40])
        print("   bad cols", cols[:20].tolist(), "... n", len(cols))
        blk = (rows // 32).unique()
        print("   bad blocks", blk[:40].tolist(), "n", len(blk))
for args in [(64, 320, 64), (17957, 320, 64), (17957, 320, 64, False, False), (17957, 320, 320), (32 * 2048, 320, 320), (32*2048, 320, 320, False, False), (32 * 2048, 160, 64), (32*2048, 960, 320, False, False)]:
    run(*args)
