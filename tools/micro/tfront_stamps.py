#!/usr/bin/env python3
"""Per-wave time split of lkgd_tattn_front (build: tools/micro/tfront_knobs.sh STAMPS -> tools/micro/libtf_STAMPS.so):
s_memtime sums of the panel prologue (token loads + LayerNorm), the wait + barrier + DMA issue at the top of each chunk, the
chunk MFMA loops and the q / k / v epilogues."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
import torch
from lkgd_amd import _lib
_lib.LIB_PATH = os.path.join(HERE, sys.argv[1] if len(sys.argv) > 1 else "libtf_STAMPS.so")
from lkgd_amd import ops
from lkgd_amd.packing import pack_tfront
DEV = "cuda:0"
B, Fr, HW, C, heads = 2, 14, 72 * 128, 320, 5
T = B * Fr * HW
x = (torch.randn(T, C, device=DEV) * 1.5).half()
w = (torch.randn(3 * C, C, device=DEV) / C ** 0.5).half()
b = torch.randn(3 * C, device=DEV) * 0.1
wf = pack_tfront(w, heads)
out = torch.empty(T, C, dtype=torch.float16, device=DEV)
for _ in range(5): ops.tattn_front(x, wf, b, out, B, Fr, HW, heads)
torch.cuda.synchronize()
d = out.view(torch.int64).reshape(-1)[: 256 * 8 * 8].reshape(256, 8, 8).double().cpu()
print("s_memtime ticks per wave, mean over workgroups; columns: prologue, sync, MFMA loops, E(q), E(k), E(v), whole kernel, panels")
for wv in range(8):
    print("  wave %d: " % wv + " ".join("%8.0f" % v for v in d[:, wv].mean(0).tolist()))
m = d.mean((0, 1))
print("  share of the wave's time: prologue %.1f %%, sync %.1f %%, MFMA loops %.1f %%, E(q) %.1f %%, E(k) %.1f %%, E(v) %.1f %%"
      % tuple(100 * m[i] / m[6] for i in range(6)))
