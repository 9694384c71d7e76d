"""Torch-tensor front end of the C-ABI (device pointers + current HIP stream).  PyTorch is plumbing here: it owns
device memory and streams; all arithmetic happens in lkgd_amd/csrc kernels.  Every function raises if a tensor is not
a CUDA(HIP) tensor of the expected dtype - there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GemmDesc, check

A_PLAIN, A_CONV3X3, A_TCONV3, A_CONV3X3_C8 = 0, 1, 2, 3

_zeros = {}

#: when a list, every GEMM launch is bracketed by HIP events on the launch stream and (start, end, algorithmic_flop)
#: is appended (bench.py's live roofline measurement); None = off
GEMM_EVENTS = None

#: lkgd_amd.replay.Plan while THIS host thread records a kernel sequence (every launch below is then also appended to it).
#: Thread-local: `pipeline.denoise` records by default since round 5, and two threads that drive two pipelines must not write
#: into each other's launch lists.  Read it as ``ops.PLAN`` (module __getattr__) or ``current_plan()``; set it with ``set_plan``.
import threading as _threading

_tls = _threading.local()


def current_plan():
    return getattr(_tls, "plan", None)


def set_plan(plan) -> None:
    _tls.plan = plan


def __getattr__(name):
    if name == "PLAN":
        return current_plan()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def _L():
    """the C-ABI library, or its recording stand-in while this thread records a plan"""
    plan = getattr(_tls, "plan", None)
    return plan.lib if plan is not None else _lib.lib()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def zeros_page(device) -> torch.Tensor:
    key = (device.type, device.index)
    z = _zeros.get(key)
    if z is None:
        z = torch.zeros(256, dtype=torch.float16, device=device)
        _zeros[key] = z
    return z


SPLITK_WORKSPACE_BYTES = 256 << 20


def splitk_workspace(device) -> torch.Tensor:
    """fp32 scratch for the split-K partial sums of few-row GEMMs; one per device and HOST THREAD: a GEMM and its reduce pass
    are two launches of one C call, stream-ordered against everything this thread enqueues - but a second host thread
    launching on the same stream could slip its own split GEMM between them (tests/thread_world.py runs ranks as threads).
    Kept in thread-local storage: a thread's 256 MB go back to the allocator when the thread ends."""
    ws = getattr(_tls, "splitk_ws", None)
    if ws is None:
        ws = _tls.splitk_ws = {}
    key = (device.type, device.index)
    w = ws.get(key)
    if w is None:
        w = torch.empty(SPLITK_WORKSPACE_BYTES // 4, dtype=torch.float32, device=device)
        ws[key] = w
    return w


def _req(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.LkgdHipError(f"{name}: expected a GPU tensor (lkgd_amd has no CPU path)")
    if t.dtype != dtype:
        raise _lib.LkgdHipError(f"{name}: expected {dtype}, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _ld(t: torch.Tensor) -> int:
    """row stride of a 2-D (possibly column-sliced) token matrix"""
    assert t.dim() == 2 and t.stride(1) == 1, "token matrices must be [T, C] with contiguous channels"
    return t.stride(0)


RowMap = Tuple[int, ...]   # (d1, m1, d2, md[, c0]): idx(row) = ((row // d1) * m1 + row % d2 + c0) % md


def rowmap_div(div: int) -> RowMap:
    """idx = row // div"""
    return (div, 1, 1, 1 << 30)


def rowmap_div_mod(div: int, mod: int) -> RowMap:
    """idx = (row // div) % mod"""
    return (div, 1, 1, mod)


def gemm(a0: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int,
         bias: Optional[torch.Tensor] = None, a1: Optional[torch.Tensor] = None, csplit: Optional[int] = None,
         mode: int = A_PLAIN, Cin: int = 0, conv: Optional[Tuple[int, int, int, int, int, int]] = None,
         tconv: Optional[Tuple[int, int]] = None, rowbias: Optional[torch.Tensor] = None,
         rowmap: Optional[RowMap] = None, res1: Optional[torch.Tensor] = None, r1: float = 1.0,
         res2: Optional[torch.Tensor] = None, r2: float = 1.0, s_acc: float = 1.0, geglu: int = 0,
         colstats: int = 0, ln: Optional[Tuple[torch.Tensor, float]] = None) -> torch.Tensor:
    """out = epilogue(A(.) @ w.T) - see include/lkgd_hip.h section 1.  ``conv`` = (Hout, Wout, Hin, Win, stride, ups);
    ``tconv`` = (F, HW).  ``colstats`` = rows per GroupNorm sample (0 = off): ``out`` feeds a GroupNorm next - where the
    tile program that will run supports it and its row blocks tile the samples, the GEMM leaves the per-(row block, channel
    pair) sums of its rounded outputs and ``out`` carries them (``out._lkgd_colstats``) to :func:`groupnorm_stats`, which
    then skips its read pass over the tensor."""
    _req(a0, torch.float16, "a0"); _req(w, torch.float16, "w"); _req(out, torch.float16, "out")
    d = GemmDesc()
    d.a0, d.w, d.out = a0.data_ptr(), w.data_ptr(), out.data_ptr()
    d.a1 = _ptr(a1)
    d.bias = _ptr(bias)
    d.zeros = zeros_page(a0.device).data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda0 = _ld(a0)
    d.lda1 = _ld(a1) if a1 is not None else 0
    d.mode, d.Cin = mode, Cin
    if mode == A_PLAIN:
        d.csplit = csplit if csplit is not None else K
    else:
        d.csplit = csplit if csplit is not None else Cin
    if conv is not None:          # (Hout, Wout, Hin, Win, stride, ups[, pad_off])
        d.Hout, d.Wout, d.Hin, d.Win, d.stride, d.ups = conv[:6]
        d.pad_off = conv[6] if len(conv) > 6 else 0
    if tconv is not None:      # (F, HW) or (F, HW, Floc, f_off) under frame sharding
        d.F, d.HW = tconv[0], tconv[1]
        d.Floc, d.f_off = (tconv[2], tconv[3]) if len(tconv) == 4 else (tconv[0], 0)
    if rowbias is not None:
        _req(rowbias, torch.float16, "rowbias")
        d.rowbias, d.ldrb = rowbias.data_ptr(), _ld(rowbias)
        d.rb_d1, d.rb_m1, d.rb_d2, d.rb_md = rowmap[:4]
        d.rb_c0 = rowmap[4] if len(rowmap) > 4 else 0
    if res1 is not None:
        _req(res1, torch.float16, "res1")
        d.res1, d.ldr1 = res1.data_ptr(), _ld(res1)
    if res2 is not None:
        _req(res2, torch.float16, "res2")
        d.res2, d.ldr2 = res2.data_ptr(), _ld(res2)
    d.ldc = _ld(out)
    d.s_acc, d.r1, d.r2 = s_acc, r1, r2
    d.geglu = int(geglu)        # 0 off, 32 / 80 = interleave width the weights were packed with
    if M < 12288:               # few-row problems may cut K into slices (lkgd_hip.h: lkgd_gemm_desc.workspace)
        ws = splitk_workspace(out.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if ln is not None:           # LayerNorm of the A rows folded into the GEMM (lkgd_gemm_desc.ln_colsum)
        _req(ln[0], torch.float32, "ln colsum")
        d.ln_colsum, d.ln_eps = ln[0].data_ptr(), float(ln[1])
    if colstats and COLSTATS and out.shape[0] == M and out.shape[1] == N:
        d.cs_rows = int(colstats)        # the tile-form choice keeps to tile rows that divide a sample
        blk = _L().lkgd_gemm_colstats_block(C.byref(d))
        if blk > 0 and int(colstats) % blk == 0:
            buf = torch.empty((M + blk - 1) // blk, N // 2, 2, dtype=torch.float32, device=out.device)   # channel pairs
            d.colstats = buf.data_ptr()
            out._lkgd_colstats = (buf, blk, out._version)      # see _stats_from_cols: void once `out` is written again
        else:
            d.cs_rows = 0                # no sums attached: the tile-form choice is not restricted either
    ev = GEMM_EVENTS
    if ev is not None:
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
        check(_L().lkgd_gemm_f16(C.byref(d), _stream()), "lkgd_gemm_f16")
        e_ev.record()
        ev.append((s_ev, e_ev, 2.0 * M * N * (72 if mode == A_CONV3X3_C8 else K)))
        return out
    check(_L().lkgd_gemm_f16(C.byref(d), _stream()), "lkgd_gemm_f16")
    return out


#: A/B switch of the GroupNorm statistics from the producing GEMM's epilogue (False = always the separate read pass)
COLSTATS = os.environ.get("LKGD_NO_COLSTATS", "0") != "1"


def _stats_from_cols(x0: torch.Tensor, x1: Optional[torch.Tensor], nsamples: int, rows_per_sample: int, eps: float,
                     as_sums: bool) -> Optional[torch.Tensor]:
    """(mean, rstd) - or raw sums - from the column sums the tensors' producing GEMMs attached (ops.gemm(colstats=True));
    None when a tensor carries none or its row blocks do not tile the samples"""
    if not COLSTATS:
        return None
    cs0 = getattr(x0, "_lkgd_colstats", None)
    cs1 = getattr(x1, "_lkgd_colstats", None) if x1 is not None else None
    if cs0 is None or (x1 is not None and cs1 is None):
        return None
    # the sums describe the tensor as its producing GEMM wrote it: any in-place torch op since then (the version counter
    # moves) voids them - the caller falls back to the read pass instead of normalising with stale statistics
    if cs0[2] != x0._version or (cs1 is not None and cs1[2] != x1._version):
        return None
    (b0, k0), (b1, k1) = cs0[:2], (cs1[:2] if cs1 is not None else (None, 1))
    if (rows_per_sample % k0 or rows_per_sample % k1 or 2 * b0.shape[1] != x0.shape[1] or
            (b1 is not None and 2 * b1.shape[1] != x1.shape[1]) or ((x0.shape[1] + (x1.shape[1] if x1 is not None else 0)) // 32) % 2):
        return None
    if b0.shape[0] * k0 < nsamples * rows_per_sample or (b1 is not None and b1.shape[0] * k1 < nsamples * rows_per_sample):
        return None
    out = torch.empty(nsamples, 32, 2, dtype=torch.float32, device=x0.device)
    check(_L().lkgd_groupnorm_stats_cols(b0.data_ptr(), k0, 2 * b0.shape[1], x0.shape[1], _ptr(b1), k1,
                                         2 * b1.shape[1] if b1 is not None else 0, x1.shape[1] if x1 is not None else 0,
                                         nsamples, rows_per_sample, eps, 1 if as_sums else 0, out.data_ptr(), _stream()),
          "lkgd_groupnorm_stats_cols")
    return out


def groupnorm_stats(x0: torch.Tensor, x1: Optional[torch.Tensor], nsamples: int, rows_per_sample: int,
                    eps: float) -> torch.Tensor:
    _req(x0, torch.float16, "x0")
    st = _stats_from_cols(x0, x1, nsamples, rows_per_sample, eps, False)
    if st is not None:
        return st
    c0 = x0.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    L = _L()
    nchunks = L.lkgd_groupnorm_chunks(rows_per_sample, c0 + c1)
    partial = torch.empty(nsamples * nchunks * 64, dtype=torch.float32, device=x0.device)
    stats = torch.empty(nsamples, 32, 2, dtype=torch.float32, device=x0.device)
    check(L.lkgd_groupnorm_stats(x0.data_ptr(), c0, _ld(x0), _ptr(x1), c1, _ld(x1) if x1 is not None else 0,
                                 nsamples, rows_per_sample, eps, partial.data_ptr(), stats.data_ptr(), _stream()),
          "lkgd_groupnorm_stats")
    return stats


def groupnorm_sums(x0: torch.Tensor, x1: Optional[torch.Tensor], nsamples: int, rows_per_sample: int) -> torch.Tensor:
    """raw fp32 (sum, sumsq) per (sample, group) of the local rows (frame-sharded GroupNorm)"""
    _req(x0, torch.float16, "x0")
    st = _stats_from_cols(x0, x1, nsamples, rows_per_sample, 0.0, True)
    if st is not None:
        return st
    c0 = x0.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    L = _L()
    nchunks = L.lkgd_groupnorm_chunks(rows_per_sample, c0 + c1)
    partial = torch.empty(nsamples * nchunks * 64, dtype=torch.float32, device=x0.device)
    sums = torch.empty(nsamples, 32, 2, dtype=torch.float32, device=x0.device)
    check(L.lkgd_groupnorm_sums(x0.data_ptr(), c0, _ld(x0), _ptr(x1), c1, _ld(x1) if x1 is not None else 0,
                                nsamples, rows_per_sample, partial.data_ptr(), sums.data_ptr(), _stream()),
          "lkgd_groupnorm_sums")
    return sums


def groupnorm_finalize(sums: torch.Tensor, count_per_group: float, eps: float) -> torch.Tensor:
    _req(sums, torch.float32, "sums")
    stats = torch.empty_like(sums)
    check(_L().lkgd_groupnorm_finalize(sums.data_ptr(), sums.shape[0], float(count_per_group), eps,
                                             stats.data_ptr(), _stream()), "lkgd_groupnorm_finalize")
    return stats


def groupnorm_finalize_parts(parts: torch.Tensor, nparts: int, part_stride: int, nsamples: int, sample_stride: int,
                              count_per_group: float, eps: float) -> torch.Tensor:
    """(mean, rstd) [nsamples, 32, 2] from the raw sums of ``nparts`` ranks inside one gathered buffer (``parts``: a float32
    view whose element r * part_stride + s * sample_stride starts rank r's sums of sample s), added in rank order"""
    _req(parts, torch.float32, "parts")
    stats = torch.empty(nsamples, 32, 2, dtype=torch.float32, device=parts.device)
    check(_L().lkgd_groupnorm_finalize_parts(parts.data_ptr(), nparts, part_stride, nsamples, sample_stride,
                                              float(count_per_group), eps, stats.data_ptr(), _stream()),
          "lkgd_groupnorm_finalize_parts")
    return stats


def groupnorm_apply(x0: torch.Tensor, x1: Optional[torch.Tensor], nsamples: int, rows_per_sample: int,
                    stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, silu: bool,
                    out: torch.Tensor) -> torch.Tensor:
    _req(x0, torch.float16, "x0"); _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    c0 = x0.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    check(_L().lkgd_groupnorm_apply(x0.data_ptr(), c0, _ld(x0), _ptr(x1), c1,
                                          _ld(x1) if x1 is not None else 0, nsamples, rows_per_sample,
                                          stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1 if silu else 0,
                                          out.data_ptr(), _ld(out), _stream()), "lkgd_groupnorm_apply")
    return out


def groupnorm_apply_segments(segments, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, silu: bool = True) -> None:
    """GroupNorm apply (+ SiLU) over a list of row segments in ONE launch: ``segments`` = [(src [rows, C], dst [rows, C], sample)],
    each normalised with ``stats[sample]``; src / dst are row-contiguous fp16 views of equal row stride (lkgd_hip.h section 2)"""
    src0 = segments[0][0]
    C_, ld = src0.shape[1], _ld(src0)
    tab = []
    for src, dst, sample in segments:
        _req(src, torch.float16, "segment src"); _req(dst, torch.float16, "segment dst")
        if src.shape != dst.shape or src.shape[1] != C_ or _ld(src) != ld or _ld(dst) != ld:
            raise _lib.LkgdHipError("groupnorm_apply_segments: segments must be [rows, C] views of one row stride")
        tab += [src.data_ptr(), dst.data_ptr(), src.shape[0], int(sample)]
    table = torch.tensor(tab, dtype=torch.int64).to(src0.device)        # (kept alive by the caller's recording, if any)
    check(_L().lkgd_groupnorm_apply_segments(table.data_ptr(), len(segments), max(s[0].shape[0] for s in segments), C_, ld,
                                             stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1 if silu else 0, _stream()),
          "lkgd_groupnorm_apply_segments")
    return table


def groupnorm_silu(x0, x1, nsamples, rows_per_sample, gamma, beta, eps, silu=True, out=None):
    C_ = x0.shape[1] + (x1.shape[1] if x1 is not None else 0)
    if out is None:
        out = torch.empty(x0.shape[0], C_, dtype=torch.float16, device=x0.device)
    _req(x0, torch.float16, "x0"); _req(gamma, torch.float32, "gamma"); _req(beta, torch.float32, "beta")
    stats = _stats_from_cols(x0, x1, nsamples, rows_per_sample, eps, False)
    if stats is not None:            # the producing GEMMs left column sums: no read pass
        return groupnorm_apply(x0, x1, nsamples, rows_per_sample, stats, gamma, beta, silu, out)
    c0 = x0.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    L = _L()
    nchunks = L.lkgd_groupnorm_chunks(rows_per_sample, c0 + c1)
    partial = torch.empty(nsamples * nchunks * 64, dtype=torch.float32, device=x0.device)
    stats = torch.empty(nsamples, 32, 2, dtype=torch.float32, device=x0.device)
    check(L.lkgd_groupnorm_silu(x0.data_ptr(), c0, _ld(x0), _ptr(x1), c1, _ld(x1) if x1 is not None else 0, nsamples,
                                rows_per_sample, eps, partial.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                1 if silu else 0, out.data_ptr(), _ld(out), _stream()), "lkgd_groupnorm_silu")
    return out


def layernorm(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float,
              out: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None,
              rowmap: Optional[RowMap] = None) -> torch.Tensor:
    """gamma = beta = None: normalise only (affine folded into the consuming Linear)"""
    _req(x, torch.float16, "x")
    if gamma is not None:
        _req(gamma, torch.float32, "gamma")
    T, C_ = x.shape
    if out is None:
        out = torch.empty(T, C_, dtype=torch.float16, device=x.device)
    d1, m1, d2, md = rowmap if rowmap is not None else (1, 0, 1, 1)
    check(_L().lkgd_layernorm(x.data_ptr(), _ld(x), T, C_, _ptr(gamma), _ptr(beta), eps,
                                    _ptr(rowbias), _ld(rowbias) if rowbias is not None else 0, d1, m1, d2, md,
                                    out.data_ptr(), _ld(out), _stream()), "lkgd_layernorm")
    return out


#: A/B switch of the one-launch feed-forward of the 72x128 level (False = LayerNorm + GEGLU GEMM + FF-out GEMM)
FF_FUSED = os.environ.get("LKGD_NO_FF_FUSED", "0") != "1"


#: A/B switch of the one-launch LayerNorm + QKV projection of the 72x128 level (False = the row-panel GEMM with the LayerNorm fold)
LN_QKV = os.environ.get("LKGD_NO_LN_QKV", "0") != "1"


def ln_qkv_ok(T: int, N: int, C_: int) -> bool:
    """the fused kernel exists for 320 -> 960 and 640 -> 1920 and pays where its 128-token panels fill the CUs: from one full
    round at C = 640 (a CFG-parallel rank's 32 256 rows: 83 us against 109 for LayerNorm + GEMM; 16 128 rows tie), from two
    at C = 320, where the alternative is the row-panel GEMM with the LayerNorm folded in (profiles/r05_ln_qkv_rows.txt)"""
    return LN_QKV and C_ in (320, 640) and N == 3 * C_ and T >= (30000 if C_ == 640 else 60000)


def ln_qkv(x: torch.Tensor, wstream: torch.Tensor, out: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """out[T, 3C] = W . LN(x) + b in one launch (lkgd_ln_qkv_c320 / _c640 by x's width)"""
    _req(x, torch.float16, "x"); _req(wstream, torch.float16, "wstream"); _req(out, torch.float16, "out")
    T, C_ = x.shape
    if C_ not in (320, 640):
        raise _lib.LkgdHipError("ln_qkv: 320 or 640 channels")
    fn = _L().lkgd_ln_qkv_c320 if C_ == 320 else _L().lkgd_ln_qkv_c640
    ev = GEMM_EVENTS
    if ev is not None:
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
    check(fn(x.data_ptr(), _ld(x), T, wstream.data_ptr(), eps, out.data_ptr(), _ld(out), _stream()), "lkgd_ln_qkv")
    if ev is not None:
        e_ev.record()
        ev.append((s_ev, e_ev, 2.0 * T * 3 * C_ * C_))
    return out


def ff_fused_ok(C_: int, inner: int) -> bool:
    return FF_FUSED and C_ == 320 and inner == 1280


def ff_fused(x: torch.Tensor, wstream: torch.Tensor, b2: torch.Tensor, out: torch.Tensor, eps: float = 1e-5,
             rowbias: Optional[torch.Tensor] = None, rowmap: Optional[RowMap] = None, s_acc: float = 1.0,
             res2: Optional[torch.Tensor] = None, r2: float = 0.0) -> torch.Tensor:
    """out = s_acc * (FF(LN(x')) + x') + r2 * res2 with x' = x + rowbias[(row // d1) % md]  (lkgd_ff_fused_c320)"""
    _req(x, torch.float16, "x"); _req(wstream, torch.float16, "wstream"); _req(b2, torch.float32, "b2")
    _req(out, torch.float16, "out")
    if x.shape[1] != 320 or out.shape != x.shape:
        raise _lib.LkgdHipError("ff_fused: x / out must be [T, 320] token matrices of the same row count")
    if res2 is not None:
        _req(res2, torch.float16, "res2")
        if res2.shape[0] != x.shape[0]:
            raise _lib.LkgdHipError("ff_fused: res2 must hold one row per token")
    if rowbias is not None:
        _req(rowbias, torch.float16, "rowbias")
    d1, m1, d2, md = (rowmap if rowmap is not None else (1, 1, 1, 1))[:4]
    if rowbias is not None and (m1 != 1 or d2 != 1 or (len(rowmap) > 4 and rowmap[4] != 0)):
        raise _lib.LkgdHipError("ff_fused: only (row // d1) % md row maps (no c0 term)")
    ev = GEMM_EVENTS        # a GEMM-family launch for bench.py's roofline line: 2 T (2560 x 320 + 320 x 1280) algorithmic FLOP
    if ev is not None:
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
    check(_L().lkgd_ff_fused_c320(x.data_ptr(), _ld(x), x.shape[0], _ptr(rowbias), _ld(rowbias) if rowbias is not None else 0,
                                  d1, min(md, 1 << 30), wstream.data_ptr(), b2.data_ptr(), eps, s_acc, _ptr(res2),
                                  _ld(res2) if res2 is not None else 0, r2, out.data_ptr(), _ld(out), _stream()),
          "lkgd_ff_fused_c320")
    if ev is not None:
        e_ev.record()
        ev.append((s_ev, e_ev, 2.0 * x.shape[0] * (2560 * 320 + 320 * 1280)))
    return out


def attn_spatial(q, k, v, out, nbatch: int, S: int, heads: int, kv_batch_map: Optional[torch.Tensor] = None,
                 scale: float = 0.125, Sq: Optional[int] = None):
    """S = key / value rows per batch entry; Sq = query rows (defaults to S; smaller on a frame-sharded DiT rank)"""
    _req(q, torch.float16, "q"); _req(k, torch.float16, "k"); _req(v, torch.float16, "v")
    if Sq is None or Sq == S:
        check(_L().lkgd_attn_spatial(q.data_ptr(), _ld(q), k.data_ptr(), _ld(k), v.data_ptr(), _ld(v),
                                     out.data_ptr(), _ld(out), nbatch, S, heads, _ptr(kv_batch_map), scale,
                                     _stream()), "lkgd_attn_spatial")
    else:
        check(_L().lkgd_attn_spatial_qk(q.data_ptr(), _ld(q), k.data_ptr(), _ld(k), v.data_ptr(), _ld(v),
                                        out.data_ptr(), _ld(out), nbatch, Sq, S, heads, _ptr(kv_batch_map), scale,
                                        _stream()), "lkgd_attn_spatial_qk")
    return out


def attn_temporal(q, k, v, out, B: int, F: int, S: int, heads: int, kv_b_map: Optional[torch.Tensor] = None,
                  scale: float = 0.125, Fq: Optional[int] = None):
    """F = key/value frames; Fq = query frames (defaults to F; smaller under frame sharding)"""
    _req(q, torch.float16, "q"); _req(k, torch.float16, "k"); _req(v, torch.float16, "v")
    check(_L().lkgd_attn_temporal(q.data_ptr(), _ld(q), k.data_ptr(), _ld(k), v.data_ptr(), _ld(v),
                                        out.data_ptr(), _ld(out), B, Fq if Fq is not None else F, F, S, heads,
                                        _ptr(kv_b_map), scale, _stream()),
          "lkgd_attn_temporal")
    return out


def tattn_front(x: torch.Tensor, wpack: torch.Tensor, bqkv: Optional[torch.Tensor], out: torch.Tensor, B: int, F: int, HW: int,
                heads: int, eps: float = 1e-5, scale: float = 0.125) -> torch.Tensor:
    """out = attn1(LayerNorm(x)) without the out-projection, fused (include/lkgd_hip.h section 5b); x / out: [B*F*HW, 320]"""
    _req(x, torch.float16, "x"); _req(wpack, torch.float16, "wpack"); _req(out, torch.float16, "out")
    check(_L().lkgd_tattn_front(x.data_ptr(), _ld(x), wpack.data_ptr(), _ptr(bqkv), out.data_ptr(), _ld(out), B, F, HW, heads,
                                eps, scale, _stream()), "lkgd_tattn_front")
    return out


def tattn_block(x: torch.Tensor, wstream: torch.Tensor, bo: torch.Tensor, out: torch.Tensor, B: int, F: int, HW: int,
                rowbias: Optional[torch.Tensor] = None, rowmap: Optional[RowMap] = None, eps: float = 1e-5) -> torch.Tensor:
    """out = to_out(attn1(LayerNorm(x))) + b_o + x (+ rowbias[rowmap(row)]): LayerNorm, Q|K|V, the attention over the F frames
    of every pixel, the out-projection and the residual in one launch (lkgd_tattn_block_c320); x / out: [B*F*HW, 320]"""
    _req(x, torch.float16, "x"); _req(wstream, torch.float16, "wstream"); _req(out, torch.float16, "out")
    _req(bo, torch.float32, "bo")
    if x.shape[0] != B * F * HW or out.shape[0] != x.shape[0] or x.shape[1] != 320 or out.shape[1] != 320:
        raise _lib.LkgdHipError("tattn_block: x / out must be [B * F * HW, 320] token matrices")
    d1, m1, d2, md = rowmap[:4] if rowmap is not None else (1, 0, 1, 1)
    c0 = rowmap[4] if rowmap is not None and len(rowmap) > 4 else 0
    if rowbias is not None:
        _req(rowbias, torch.float16, "rowbias")
    nflop = 2 * x.shape[0] * (960 * 320 + 320 * 320) + 4 * x.shape[0] * 16 * 320
    ev = GEMM_EVENTS        # a GEMM-family launch for bench.py's roofline line (projections, out-projection, 16 x 16 attention)
    if ev is not None:
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
    check(_L().lkgd_tattn_block_c320(x.data_ptr(), _ld(x), wstream.data_ptr(), bo.data_ptr(), _ptr(rowbias),
                                     _ld(rowbias) if rowbias is not None else 0, d1, m1, d2, min(md, 0x7fffffff), c0,
                                     out.data_ptr(), _ld(out), B, F, HW, eps, _stream()), "lkgd_tattn_block_c320")
    if ev is not None:
        e_ev.record()
        ev.append((s_ev, e_ev, float(nflop)))
    return out


def tattn_block_ok(C_: int, heads: int, F: int, HW: int) -> bool:
    return TBLOCK and C_ == 320 and heads == 5 and 1 <= F <= 16


#: A/B switch of the one-launch temporal attention (False = fused front + out-projection GEMM)
TBLOCK = os.environ.get("LKGD_NO_TBLOCK", "0") != "1"


def tattn_front_ok(C_: int, heads: int, F: int, HW: int) -> bool:
    return TFRONT and C_ == 320 and heads * 64 == C_ and 1 <= F <= 16 and HW % 16 == 0


#: A/B switch of the fused temporal-attention front (False = LayerNorm, QKV GEMM and attention kernel as three launches)
TFRONT = os.environ.get("LKGD_NO_TFRONT", "0") != "1"


def attn_cross(q, k, v, out, heads: int, ncontexts: int, Lk: int, rowmap: RowMap, scale: float = 0.125):
    """rows of q against the Lk keys of context rowmap(row); k / v: [ncontexts * Lk, heads * 64] (lkgd_hip.h section 15)"""
    _req(q, torch.float16, "q"); _req(k, torch.float16, "k"); _req(v, torch.float16, "v"); _req(out, torch.float16, "out")
    d1, m1, d2, md = rowmap[:4]
    c0 = rowmap[4] if len(rowmap) > 4 else 0
    check(_L().lkgd_attn_cross(q.data_ptr(), _ld(q), k.data_ptr(), _ld(k), v.data_ptr(), _ld(v), out.data_ptr(), _ld(out),
                               q.shape[0], heads, ncontexts, Lk, d1, m1, d2, md, c0, scale, _stream()), "lkgd_attn_cross")
    return out


#: A/B switch of the LayerNorm fold into the row-panel QKV projection (False = LayerNorm kernel + GEMM)
LNFOLD = os.environ.get("LKGD_NO_LNFOLD", "0") != "1"


def gemm_ln_ok(M: int, N: int, K: int) -> bool:
    """shapes for which gemm(..., ln=...) runs (the row-panel program) AND pays: K = LayerNorm width <= 320 and enough 256-row
    panels to fill the CUs (a rank of 8 has 144 at the 72x128 level: LayerNorm + 256x320 tiles stay ahead there)"""
    return LNFOLD and K <= 320 and K % 64 == 0 and K >= 192 and M >= 60000 and N % 8 == 0


def attn_dense(q, k, v, out, nbatch: int, S: int, heads: int, head_dim: int, scale: Optional[float] = None):
    """short-sequence attention with any head_dim <= 128 (lkgd_hip.h section 16): q, k, v, out [nbatch*S, heads*head_dim]"""
    _req(q, torch.float16, "q"); _req(k, torch.float16, "k"); _req(v, torch.float16, "v"); _req(out, torch.float16, "out")
    if scale is None:
        scale = head_dim ** -0.5
    check(_L().lkgd_attn_dense(q.data_ptr(), _ld(q), k.data_ptr(), _ld(k), v.data_ptr(), _ld(v), out.data_ptr(), _ld(out),
                               nbatch, S, heads, head_dim, scale, _stream()), "lkgd_attn_dense")
    return out


def prepare_unet_input(latents: torch.Tensor, image_latents: torch.Tensor, cfg: int, sigma: float,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B,F,4,H,W] latents (+ [cfg*B,F,4,H,W] image latents) -> channels-last tokens [cfg*B*F*H*W, 8]"""
    B, F, _, H, W = latents.shape
    _req(image_latents, torch.float16, "image_latents")
    if not latents.is_cuda or latents.dtype not in (torch.float16, torch.float32):
        raise _lib.LkgdHipError("latents must be a GPU fp16/fp32 tensor")
    assert latents.is_contiguous() and image_latents.is_contiguous()
    assert image_latents.shape == (cfg * B, F, 4, H, W), image_latents.shape
    if out is None:
        out = torch.empty(cfg * B * F * H * W, 8, dtype=torch.float16, device=latents.device)
    check(_L().lkgd_prepare_unet_input(latents.data_ptr(), int(latents.dtype == torch.float32),
                                             image_latents.data_ptr(), B, F, H, W, cfg, sigma, out.data_ptr(),
                                             _stream()), "lkgd_prepare_unet_input")
    return out


def cfg_euler_step(noise_tokens: torch.Tensor, latents: torch.Tensor, guidance: Optional[torch.Tensor], cfg: int,
                   sigma: float, sigma_next: float, v_prediction: bool = True) -> torch.Tensor:
    """in-place Euler update of ``latents`` [B,F,4,H,W] from channels-last noise tokens [cfg*B*F*H*W, 4]"""
    B, F, _, H, W = latents.shape
    _req(noise_tokens, torch.float16, "noise_tokens")
    assert latents.is_contiguous() and noise_tokens.is_contiguous()
    check(_L().lkgd_cfg_euler_step(noise_tokens.data_ptr(), latents.data_ptr(),
                                         int(latents.dtype == torch.float32), _ptr(guidance), B, F, H, W, cfg, sigma,
                                         sigma_next, 1 if v_prediction else 0, _stream()), "lkgd_cfg_euler_step")
    return latents


def shard_rows(src: torch.Tensor, dst: torch.Tensor, fl: int, HW: int, C_: int, px, pack: bool) -> torch.Tensor:
    """pack: src [fl, HW, C] -> dst rows grouped by destination pixel shard (px = pixels per shard); not pack: the inverse
    (lkgd_shard_rows).  Both contiguous fp16."""
    _req(src, torch.float16, "src"); _req(dst, torch.float16, "dst")
    if not (src.is_contiguous() and dst.is_contiguous()) or src.numel() != fl * HW * C_ or dst.numel() != src.numel():
        raise _lib.LkgdHipError("shard_rows: contiguous [fl * HW, C] buffers expected")
    import ctypes
    tab = (ctypes.c_int32 * len(px))(*px)
    # (always the real library: this runs inside the host-side exchange steps, which a recorded plan re-runs as a whole -
    # recording the launch as well would run it twice per replay)
    check(_lib.lib().lkgd_shard_rows(src.data_ptr(), dst.data_ptr(), fl, HW, C_, len(px), ctypes.cast(tab, ctypes.c_void_p),
                                     int(pack), _stream()), "lkgd_shard_rows")
    return dst


def tokens_to_nchw(tokens: torch.Tensor, N: int, C_: int, H: int, W: int) -> torch.Tensor:
    _req(tokens, torch.float16, "tokens")
    out = torch.empty(N, C_, H, W, dtype=torch.float16, device=tokens.device)
    check(_L().lkgd_tokens_to_nchw(tokens.data_ptr(), _ld(tokens), N, C_, H * W, out.data_ptr(), _stream()),
          "lkgd_tokens_to_nchw")
    return out


def nchw_to_tokens(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x, torch.float16, "x")
    N, C_, H, W = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(N * H * W, C_, dtype=torch.float16, device=x.device)
    check(_L().lkgd_nchw_to_tokens(x.data_ptr(), N, C_, H * W, out.data_ptr(), _ld(out), _stream()),
          "lkgd_nchw_to_tokens")
    return out


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    _req(t, torch.float32, "t")
    out = torch.empty(t.numel(), dim, dtype=torch.float16, device=t.device)
    check(_L().lkgd_timestep_embedding(t.data_ptr(), t.numel(), dim, out.data_ptr(), dim, _stream()),
          "lkgd_timestep_embedding")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.float16, "x")
    y = torch.empty_like(x)
    check(_L().lkgd_silu(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "lkgd_silu")
    return y


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _req(a, torch.float16, "a"); _req(b, torch.float16, "b")
    y = torch.empty_like(a)
    check(_L().lkgd_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), "lkgd_add")
    return y


def fsm_rows(a: torch.Tensor, out: torch.Tensor, *, pairs: int, HW: int, C_: int, a_rows: Tuple[int, int],
             o_rows: Tuple[int, int], res: Optional[torch.Tensor] = None, r_rows: Tuple[int, int] = (0, 0),
             bias: Optional[torch.Tensor] = None, bias_map: Tuple[int, int, int] = (1, 0, 1),
             csr: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """FSM hook row kernel (include/lkgd_hip.h section 9).  ``x_rows`` = (rows per pair, row offset); ``csr`` =
    (csr_off int32 [pairs*HW+1], csr_pt int32 [pairs*P], gather_idx int32 [pairs*P], vis fp32 [pairs*P])."""
    _req(a, torch.float16, "a"); _req(out, torch.float16, "out")
    d = _lib.FsmDesc()
    d.a, d.out, d.lda, d.ldo = a.data_ptr(), out.data_ptr(), _ld(a), _ld(out)
    d.a_pair_rows, d.a_off = a_rows
    d.o_pair_rows, d.o_off = o_rows
    d.pairs, d.HW, d.C = pairs, HW, C_
    if res is not None:
        _req(res, torch.float16, "res")
        d.res, d.ldr = res.data_ptr(), _ld(res)
        d.r_pair_rows, d.r_off = r_rows
    if bias is not None:
        _req(bias, torch.float16, "bias")
        d.bias, d.ldb = bias.data_ptr(), _ld(bias)
        d.bias_mul, d.bias_add, d.bias_div = bias_map
    if csr is not None:
        off, pt, gi, vis = csr
        for t, dt, n in ((off, torch.int32, "csr_off"), (pt, torch.int32, "csr_pt"), (gi, torch.int32, "gather_idx"),
                         (vis, torch.float32, "vis")):
            if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                raise _lib.LkgdHipError(f"{n} must be a contiguous GPU {dt} tensor")
        if off.numel() != pairs * HW + 1 or pt.numel() != gi.numel() or pt.numel() != vis.numel():
            raise _lib.LkgdHipError("fsm_rows: CSR table sizes do not match pairs*HW / pairs*P")
        d.csr_off, d.csr_pt, d.gather_idx, d.vis = off.data_ptr(), pt.data_ptr(), gi.data_ptr(), vis.data_ptr()
        d.P = pt.numel() // pairs
    check(_L().lkgd_fsm_rows(C.byref(d), _stream()), "lkgd_fsm_rows")
    return out


def conv3x3_small(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], nimg: int, Hin: int, Win: int,
                  stride: int = 1, silu: bool = True) -> torch.Tensor:
    """direct 3x3 convolution for small channel counts (include/lkgd_hip.h section 10); x [nimg*Hin*Win, Cin] tokens,
    w [Cout, 3, 3, Cin] fp16 -> [nimg*Hout*Wout, Cout]"""
    _req(x, torch.float16, "x"); _req(w, torch.float16, "w")
    cout, cin = w.shape[0], w.shape[3]
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    out = torch.empty(nimg * Hout * Wout, cout, dtype=torch.float16, device=x.device)
    check(_L().lkgd_conv3x3_small(x.data_ptr(), cin, _ld(x), w.data_ptr(), _ptr(bias), out.data_ptr(), cout,
                                        _ld(out), nimg, Hin, Win, stride, int(silu), _stream()), "lkgd_conv3x3_small")
    return out


def softmax_rows(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """row softmax of an fp16 score matrix (include/lkgd_hip.h section 12); in place when out is None"""
    _req(x, torch.float16, "x")
    out = x if out is None else out
    check(_L().lkgd_softmax_rows(x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), x.shape[0], x.shape[1], _stream()),
          "lkgd_softmax_rows")
    return out


def time_conv_out(tokens: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, nbatch: int, F: int, H: int, W: int,
                  dtype=torch.float32) -> torch.Tensor:
    """Conv3d (3,1,1) over the frames on 3 channels + channels-last -> NCHW (section 12); w fp32 [3, 3, 3] = (co, ci, kt)"""
    _req(tokens, torch.float16, "tokens"); _req(w, torch.float32, "w"); _req(bias, torch.float32, "bias")
    out = torch.empty(nbatch * F, 3, H, W, dtype=dtype, device=tokens.device)
    check(_L().lkgd_time_conv_out(tokens.data_ptr(), _ld(tokens), w.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                  int(dtype == torch.float32), nbatch, F, H * W, _stream()), "lkgd_time_conv_out")
    return out


def vit_patchify(x: torch.Tensor, size: int, patch: int) -> torch.Tensor:
    """bilinear resize to size x size (align_corners=False) + patch unfold (include/lkgd_hip.h section 13):
    fp32 [N, C, H, W] -> fp16 [N*(size/patch)^2, C*patch*patch]"""
    _req(x, torch.float32, "x")
    x = x.contiguous()
    N, C_, H, W = x.shape
    g = size // patch
    out = torch.empty(N * g * g, C_ * patch * patch, dtype=torch.float16, device=x.device)
    check(_L().lkgd_vit_patchify(x.data_ptr(), N, C_, H, W, out.data_ptr(), size, patch, _stream()), "lkgd_vit_patchify")
    return out


def gelu_tanh_(x: torch.Tensor) -> torch.Tensor:
    """F.gelu(x, approximate="tanh") in place (include/lkgd_hip.h section 14)"""
    _req(x, torch.float16, "x")
    if not x.is_contiguous():
        raise _lib.LkgdHipError("gelu_tanh_ needs a contiguous tensor")
    check(_L().lkgd_gelu_tanh(x.data_ptr(), x.data_ptr(), x.numel(), _stream()), "lkgd_gelu_tanh")
    return x


def gated_add(x: torch.Tensor, gate: torch.Tensor, res: torch.Tensor, rows_per_batch: int, split: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = res + gate[(row // rows_per_batch) * 2 + (row % rows_per_batch >= split)] * x  (section 14); gate fp32 [2*B, C]"""
    _req(x, torch.float16, "x"); _req(res, torch.float16, "res"); _req(gate, torch.float32, "gate")
    if out is None:
        out = torch.empty_like(res)
    check(_L().lkgd_gated_add(x.data_ptr(), _ld(x), gate.data_ptr(), res.data_ptr(), _ld(res), out.data_ptr(), _ld(out),
                              x.shape[0], x.shape[1], rows_per_batch, split, _stream()), "lkgd_gated_add")
    return out


def scale(x: torch.Tensor, s: float) -> torch.Tensor:
    _req(x, torch.float16, "x")
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_L().lkgd_scale(x.data_ptr(), y.data_ptr(), x.numel(), s, _stream()), "lkgd_scale")
    return y


def euler_step(model_output: torch.Tensor, sample: torch.Tensor, sigma: float, sigma_next: float,
               v_prediction: bool = True, noise: torch.Tensor = None, sigma_hat: float = None, s_noise: float = 1.0,
               churn: float = 0.0) -> torch.Tensor:
    """one Euler step; with `noise` (fp16, same shape) the stochastic form: sample += fp16(fp16(noise * s_noise) * churn),
    then the step from sigma_hat to sigma_next (lkgd_hip.h section 8)"""
    _req(model_output, torch.float16, "model_output")
    if not sample.is_cuda or sample.dtype not in (torch.float16, torch.float32):
        raise _lib.LkgdHipError("sample must be a GPU fp16/fp32 tensor")
    mo, sm = model_output.contiguous(), sample.contiguous()
    prev = torch.empty_like(mo)
    if noise is None:
        check(_L().lkgd_euler_step(mo.data_ptr(), sm.data_ptr(), int(sm.dtype == torch.float32), prev.data_ptr(),
                                   mo.numel(), sigma, sigma_next, 1 if v_prediction else 0, _stream()),
              "lkgd_euler_step")
        return prev
    _req(noise, torch.float16, "noise")
    if noise.numel() != mo.numel():
        raise _lib.LkgdHipError("noise must have model_output's shape")
    nz = noise.contiguous()
    check(_L().lkgd_euler_step_churn(mo.data_ptr(), sm.data_ptr(), int(sm.dtype == torch.float32), nz.data_ptr(),
                                     prev.data_ptr(), mo.numel(), sigma, sigma_hat, s_noise, churn, sigma_next,
                                     1 if v_prediction else 0, _stream()), "lkgd_euler_step_churn")
    return prev


def conv1d_reflect(x: torch.Tensor, taps: torch.Tensor, axis: int) -> torch.Tensor:
    """fp32 [B,C,H,W]: 1-D filter along W (axis 1) or H (axis 0) with reflect padding (lkgd_hip.h section 11)"""
    _req(x, torch.float32, "x"); _req(taps, torch.float32, "taps")
    assert x.is_contiguous() and taps.is_contiguous() and x.dim() == 4
    b, c, h, w = x.shape
    out = torch.empty_like(x)
    check(_L().lkgd_conv1d_reflect(x.data_ptr(), out.data_ptr(), b * c, h, w, taps.data_ptr(), taps.numel(), axis,
                                   _stream()), "lkgd_conv1d_reflect")
    return out


def resize_bicubic_ac(x: torch.Tensor, ho: int, wo: int) -> torch.Tensor:
    """fp32 [B,C,H,W] -> [B,C,ho,wo], bicubic (A = -0.75), align_corners=True"""
    _req(x, torch.float32, "x")
    assert x.is_contiguous() and x.dim() == 4
    b, c, h, w = x.shape
    out = torch.empty(b, c, ho, wo, dtype=torch.float32, device=x.device)
    check(_L().lkgd_resize_bicubic_ac(x.data_ptr(), b * c, h, w, out.data_ptr(), ho, wo, _stream()),
          "lkgd_resize_bicubic_ac")
    return out
