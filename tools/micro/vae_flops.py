import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lkgd_amd import ops, unet as pu, vae as pv
tot = [0.0, 0]
_g = ops.gemm
def g(a0, w, out, *, M, N, K, **kw):
    tot[0] += 2.0 * M * N * K; tot[1] += 1
    return _g(a0, w, out, M=M, N=N, K=K, **kw)
ops.gemm = g
for mod in list(sys.modules.values()):
    if mod and getattr(mod, "__name__", "").startswith("lkgd_amd") and hasattr(mod, "ops") and mod.ops is ops: pass
dev = torch.device("cuda:0")
with torch.device("meta"):
    v = pv.AutoencoderKLTemporalDecoder()
v = v.to(torch.float16).to_empty(device=dev)
pu.init_synthetic_weights_(v, seed=2)
z = torch.randn(14, 4, 72, 128).half().to(dev)
out = v.decode(z, num_frames=14).sample
torch.cuda.synchronize()
print("decode GEMM TFLOP", tot[0] / 1e12, "launches", tot[1])
