"""Host side of the SVD / LKGD UNet for MI355X: the reference's module tree and ``forward`` signatures, with every
tensor op of the forward executed by the gfx950 kernels of ``lkgd_amd/csrc`` (through ``lkgd_amd.ops``).

Mirrors
* /root/reference/models/unet_spatio_temporal_condition_controlnet.py:32,70-245,358-508
  (``UNetSpatioTemporalConditionControlNetModel`` - stock signature), and
* /root/reference/models/unet_spatio_temporal_condition.py:34,197-225,448-693
  (``UNetSpatioTemporalConditionModel`` - LKGD signature, latent-knowledge fuse :536-595),
and the diffusers==0.27.2 blocks those files instantiate.  Module and parameter names equal
diffusers' so ``load_state_dict`` of a real SVD ``unet`` checkpoint works unchanged; block classes are *named*
``BasicTransformerBlock`` / ``TemporalBasicTransformerBlock`` and expose the attributes the reference's ``patch``
module pokes.

MI355X-first design decisions:
* activations are channels-last fp16 token matrices [N*H*W, C] end to end - the reference's NCHW<->[N,HW,C]<->
  [B*HW,F,C] permutes are index maps inside kernels, never copies;
* nn.Conv*/Linear/Norm layers here are PARAMETER HOLDERS (state-dict compatibility); they are never called.
  ``prepare()`` re-lays the weights for the kernels once (K-contiguous fp16, fused QKV, tile-interleaved GEGLU,
  all time_emb_proj of the model concatenated into one GEMM, cross-attention value path folded to one matrix);
* the CLIP cross-attention has exactly one key/value token, so softmax == 1 and attn2 reduces to the row bias
  to_out(to_v(e)) - computed for all 32 attn2 layers by ONE small GEMM per forward and folded into the epilogue of
  the preceding projection (SURVEY.md finding 5);
* residual adds, time-embedding adds, GEGLU, AlphaBlender mixes are GEMM epilogues; skip concats and nearest-2x
  upsampling are folded into the consumers' gathers.
No CPU / eager fallback: all ops raise off-GPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops
from . import replay as _replay
from . import trace as _trace
from ._lib import LkgdHipError
from .packing import pack_conv3x3, pack_conv3x3_c8, pack_ff_fused, pack_geglu, pack_linear, pack_tconv3

#: row order of the temporal cross-attention context, see SURVEY.md App. C11.  "interleaved_0_27" reproduces
#: diffusers 0.27.x (rows (pixel, batch)-ordered against (batch, pixel)-ordered hidden rows); "batch_major" = later.
TIME_CONTEXT_ORDER = "interleaved_0_27"

#: GroupNorm eps of the res blocks per container (kept in ONE table, see oracle/blocks.py RESNET_EPS)
RESNET_EPS = {
    "CrossAttnDownBlockSpatioTemporal": 1e-6,
    "DownBlockSpatioTemporal": 1e-5,
    "UNetMidBlockSpatioTemporal": 1e-5,
    "UpBlockSpatioTemporal": 1e-6,
    "CrossAttnUpBlockSpatioTemporal": 1e-6,
}


@dataclass
class UNetConfig:
    """constructor keywords of the reference classes (unet_..._controlnet.py:70-96)"""
    sample_size: Optional[int] = 96
    in_channels: int = 8
    out_channels: int = 4
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                                         "CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal")
    up_block_types: Tuple[str, ...] = ("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal",
                                       "CrossAttnUpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal")
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 768
    layers_per_block: int = 2
    cross_attention_dim: int = 1024
    transformer_layers_per_block: int = 1
    num_attention_heads: Tuple[int, ...] = (5, 10, 20, 20)
    num_frames: int = 14


@dataclass
class UNetSpatioTemporalConditionOutput:
    sample: torch.Tensor = None


# ----------------------------------------------------------------------------------------------- execution context
class Ctx:
    """per-forward state: geometry, time-embedding table, cross-attention bias tables"""

    def __init__(self, B: int, F: int, H: int, W: int, device, shard=None):
        self.B, self.F, self.H, self.W = B, F, H, W      # LOCAL batch entries / frames
        self.device = device
        # frame / CFG sharding over GPUs (lkgd_amd/dist_run.py); unsharded: totals == locals, offsets 0
        self.shard = shard
        self.F_total = shard.F_total if shard is not None else F
        self.f0 = shard.f0 if shard is not None else 0
        self.B_total = shard.B_total if shard is not None else B
        self.b0 = shard.b0 if shard is not None else 0
        self.temb_all: Optional[torch.Tensor] = None      # [B, sum C] fp16
        self.xb_all: Optional[torch.Tensor] = None        # [B, sum C] fp16
        self.joint_blocks = None
        self.spatial_partner: Optional[torch.Tensor] = None   # joint attention maps (patch API)
        self.temporal_partner: Optional[torch.Tensor] = None
        self.entry_partner: Optional[List[int]] = None
        self.lora = None               # lkgd_amd.lora.EntryPlan when the model carries LoRA wrappers (masked LoRA forward)
        self.xb_runs: Optional[List[torch.Tensor]] = None     # cross-attention bias tables per entry run (LoRA on attn2)
        # multi-token context (Lk > 1): the literal attn2 path (_cross_literal); xb_all is then a table of zeros
        self.cross_Lk = 1
        self.cross_e: Optional[torch.Tensor] = None       # [B_total * Lk, cross_attention_dim] fp16
        self.flip_mirror = False       # flip=True joint attention on a frame-sharded rank: K | V of attn1n come from the mirror shard

    @property
    def frames_sharded(self) -> bool:
        return self.F_total != self.F

    @property
    def N(self):
        return self.B * self.F

    @property
    def HW(self):
        return self.H * self.W

    @property
    def T(self):
        return self.B * self.F * self.H * self.W

    def new(self, rows: int, cols: int) -> torch.Tensor:
        return torch.empty(rows, cols, dtype=torch.float16, device=self.device)


def _f32(p: torch.Tensor) -> torch.Tensor:
    return p.detach().to(torch.float32).contiguous()


class _Packed:
    """marker base: modules that own kernel-layout copies of their weights"""
    _pk = None

    def pack(self, model: "_UNetBase"):
        raise NotImplementedError


# ----------------------------------------------------------------------------------------------- embeddings
class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int, out_dim: Optional[int] = None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def pack(self):
        self._pk = SimpleNamespace(w1=pack_linear(self.linear_1.weight), b1=_f32(self.linear_1.bias),
                                   w2=pack_linear(self.linear_2.weight), b2=_f32(self.linear_2.bias))

    def run(self, x: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x fp16 [M, in] -> linear_2(silu(linear_1(x))) (+ res)"""
        pk = self._pk
        M = x.shape[0]
        h = torch.empty(M, pk.w1.shape[0], dtype=torch.float16, device=x.device)
        ops.gemm(x, pk.w1, h, M=M, N=pk.w1.shape[0], K=pk.w1.shape[1], bias=pk.b1)
        h = ops.silu(h)
        out = torch.empty(M, pk.w2.shape[0], dtype=torch.float16, device=x.device)
        ops.gemm(h, pk.w2, out, M=M, N=pk.w2.shape[0], K=pk.w2.shape[1], bias=pk.b2, res1=res)
        return out


# ----------------------------------------------------------------------------------------------- attention holders
class Attention(nn.Module):
    """parameter holder for diffusers Attention (to_q/k/v without bias, to_out.0 with bias); head_dim must be 64"""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int], heads: int, dim_head: int):
        super().__init__()
        if dim_head != 64:
            raise LkgdHipError(f"lkgd_amd attention kernels are built for head_dim 64 (got {dim_head})")
        self.inner_dim = heads * dim_head
        self.heads = heads
        self.out_dim = query_dim
        self.scale = dim_head ** -0.5
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.to_k = nn.Linear(kv, self.inner_dim, bias=False)
        self.to_v = nn.Linear(kv, self.inner_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=True), nn.Dropout(0.0)])

    def pack_self(self, norm: Optional[nn.LayerNorm] = None, adapters=((), (), (), ())):
        """fused QKV weight; with `norm` the preceding LayerNorm's affine is folded in (QKV then carries a bias).
        `adapters` = the LoRA adapter names folded into (to_q, to_k, to_v, to_out.0) for this variant (lkgd_amd/lora.py)"""
        aq, ak, av, ao = adapters
        w = torch.cat([_eff_weight(self.to_q, aq), _eff_weight(self.to_k, ak), _eff_weight(self.to_v, av)], dim=0)
        bqkv = None
        if norm is not None:
            w, bqkv = _fold_ln(norm, w, None)
            bqkv = bqkv.contiguous()
        wq = pack_linear(w)
        # column sums of the PACKED (fp16) weights: what the LayerNorm fold of ops.gemm(ln=...) subtracts, times the row mean
        cs = wq.float().sum(dim=1).contiguous() if norm is not None else None
        return SimpleNamespace(wqkv=wq, bqkv=bqkv, cs=cs,
                               wo=pack_linear(_eff_weight(self.to_out[0], ao)), bo=_f32(self.to_out[0].bias))

    def pack_cross(self, norm: nn.LayerNorm):
        """literal cross-attention (context of more than one token): Q projection with the preceding LayerNorm folded in,
        fused K|V projection of the context, out-projection"""
        wq, bq = _fold_ln(norm, _eff_weight(self.to_q), None)
        wkv = torch.cat([_eff_weight(self.to_k), _eff_weight(self.to_v)], dim=0)
        return SimpleNamespace(wq=pack_linear(wq), bq=bq.contiguous(), wkv=pack_linear(wkv),
                               wo=pack_linear(_eff_weight(self.to_out[0])), bo=_f32(self.to_out[0].bias))

    def fold_cross(self, adapters=((), ())):
        """single key/value token => attn2(x, e) == to_out(to_v(e)): returns (W_o @ W_v [C,1024] fp32, b_o); LoRA on
        to_q / to_k of a one-token attention is dead (softmax over one key)"""
        av, ao = adapters
        wo = _eff_weight(self.to_out[0], ao).to(torch.float32)
        wv = _eff_weight(self.to_v, av).to(torch.float32)
        return wo @ wv, _f32(self.to_out[0].bias)


def _eff_weight(lin, adapters=()) -> torch.Tensor:
    """weight of a projection with the named LoRA adapters folded in (fp32 when the layer is wrapped)"""
    if hasattr(lin, "effective_weight"):
        return lin.effective_weight(adapters)
    return lin.weight.detach()


def _gemm_runs(ctx: "Ctx", a: torch.Tensor, out: torch.Tensor, pick, *, N: int, K: int, rowmap=None, res1=None,
               res2=None, rows: Optional[int] = None, **kw) -> None:
    """masked-LoRA form of a projection over all T rows: one launch per run of consecutive batch entries that share a
    weight variant (lkgd_amd/lora.py).  pick(i) -> (w, bias, rowbias table or None) of run i; a row map is shifted to the
    run's first row (idx(m) = ((m / d1) * m1 + m % d2 + c0) % md with the run start a multiple of d1 and d2).  ``rows``: token
    rows per batch entry in the layout of ``a`` (default: the rank's frames x pixels; the pixel-re-sharded layout of a
    frame-sharded temporal block has all F frames of a pixel slice per entry)"""
    rows = ctx.F * ctx.HW if rows is None else rows
    for i, (b0, b1) in enumerate(ctx.lora.runs):
        r0, r1 = b0 * rows, b1 * rows
        w, bias, rb = pick(i)
        rm = None
        if rb is not None:
            if r0 % rowmap[0] or r0 % rowmap[2]:
                raise LkgdHipError("internal: entry run does not start on a row-map period")
            rm = (rowmap[0], rowmap[1], rowmap[2], rowmap[3],
                  (rowmap[4] if len(rowmap) > 4 else 0) + (r0 // rowmap[0]) * rowmap[1])
        ops.gemm(a[r0:r1], w, out[r0:r1], M=r1 - r0, N=N, K=K, bias=bias, rowbias=rb, rowmap=rm,
                 res1=res1[r0:r1] if res1 is not None else None, res2=res2[r0:r1] if res2 is not None else None, **kw)


def _attn_variant(block, which: str, ctx: "Ctx", run: int, spatial: bool):
    """packed weights of block.attn1 / block.attn1n for entry run `run` (built on demand per adapter subset, cached on
    the block's pack).  K / V of the joint attention see the PARTNER entry's adapters: their input rows are consumed by
    the partner (patch/patch.py:466-468,:889-892)."""
    pk = block._pk
    attn = block.attn1 if which == "a1" else block.attn1n
    P = ctx.lora
    jn = which == "a1n"
    key = (which, P.adapters(attn.to_q, run), P.adapters(attn.to_k, run, jn), P.adapters(attn.to_v, run, jn),
           P.adapters(attn.to_out[0], run))
    v = pk.var.get(key)
    if v is None:
        with _replay.invariant():
            v = attn.pack_self(block.norm1, key[1:])
            if jn:
                _pack_joint_post(block, v, spatial, key[4])
        pk.var[key] = v
    return v


def _pack_joint_post(block, pk, spatial: bool, out_adapters=()):
    """fold the joint branch's post-processing (patch/patch.py:484-494) into attn1n's out-projection:
    conv:      conv1n(to_out(a))  = a @ (Wc Wo)^T + Wc bo
    scale:     scale1n * to_out(a) = a @ (diag(s) Wo)^T + s * bo
    conv_fuse: cat(o[mask], o[~mask]) @ Wc^T, chunked back = own/partner halves of Wc times Wo (K = 2C, spatial only;
               the temporal branch applies no post for 'conv_fuse', patch.py:647-650)"""
    post = getattr(block, "post", "conv")
    wo = _eff_weight(block.attn1n.to_out[0], out_adapters).to(torch.float32)
    bo = block.attn1n.to_out[0].bias.detach().to(torch.float32)
    C_ = wo.shape[0]
    if post == "conv":
        wc = block.conv1n.weight.detach().to(torch.float32)
        pk.jw, pk.jb = pack_linear(wc @ wo), (wc @ bo).contiguous()
    elif post == "scale":
        sc = block.scale1n.detach().to(torch.float32).reshape(-1)
        pk.jw, pk.jb = pack_linear(sc[:, None] * wo), (sc * bo).contiguous()
    elif post == "conv_fuse" and spatial:
        wc = block.conv1n.weight.detach().to(torch.float32)
        own_m, par_m = wc[:C_, :C_], wc[:C_, C_:]          # masked entries: fx = o_m A^T + o_u B^T
        own_u, par_u = wc[C_:, C_:], wc[C_:, :C_]          # unmasked:       fy = o_u D^T + o_m C^T
        pk.jw_m = pack_linear(torch.cat([own_m @ wo, par_m @ wo], dim=1))
        pk.jb_m = ((own_m + par_m) @ bo).contiguous()
        pk.jw_u = pack_linear(torch.cat([own_u @ wo, par_u @ wo], dim=1))
        pk.jb_u = ((own_u + par_u) @ bo).contiguous()
    elif post == "conv_fuse":
        pk.jw, pk.jb = pack_linear(wo), bo.contiguous()
    else:
        raise LkgdHipError(f"joint attention: unknown post '{post}'")
    pk.post = post


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim_out or dim)])

    def pack(self, norm: Optional[nn.LayerNorm] = None):
        w, b = self.net[0].proj.weight.detach(), self.net[0].proj.bias.detach()
        if norm is not None:
            w, b = _fold_ln(norm, w, b)
        wp, bp, half = pack_geglu(w, b)
        pk = SimpleNamespace(wp=wp, bp=bp, half=half, wo=pack_linear(self.net[2].weight), bo=_f32(self.net[2].bias),
                             wfused=None)
        w2 = self.net[2].weight.detach()
        if norm is not None and ops.ff_fused_ok(w.shape[1], w2.shape[1]) and w2.shape[0] == w.shape[1]:
            # the 72x128 level: LayerNorm + GEGLU + FF-out + residual(s) in one launch (lkgd_ff_fused_c320)
            pk.wfused = pack_ff_fused(w.float(), b.float(), w2.float())
        return pk


def _fold_ln(norm: nn.LayerNorm, w: torch.Tensor, b: Optional[torch.Tensor]):
    """LayerNorm affine folded into the Linear that consumes it (exact algebra, done once at pack time):
    (z*gamma + beta) W^T + b  ==  z (W*diag(gamma))^T + (b + W beta)  with z the plain normalised input"""
    g, be = norm.weight.detach().float(), norm.bias.detach().float()
    w32 = w.detach().float()
    b32 = w32 @ be + (b.detach().float() if b is not None else 0.0)
    return w32 * g[None, :], b32


def _ff_ln(ctx: Ctx, pk, x: torch.Tensor, rowbias=None, rowmap=None, s_acc: float = 1.0, res2=None, r2: float = 0.0):
    """s_acc * (FF(LN(x')) + x') + r2 * res2 with x' = x + rowbias[rowmap(row)]: norm3 -> ff / norm_in -> ff_in with their
    residuals.  One launch where the fused kernel exists (C = 320), LayerNorm + two GEMMs elsewhere."""
    if getattr(pk, "wfused", None) is not None and ops.FF_FUSED:
        return ops.ff_fused(x, pk.wfused, pk.bo, ctx.new(x.shape[0], x.shape[1]), 1e-5, rowbias, rowmap, s_acc, res2, r2)
    ln = ops.layernorm(x, None, None, 1e-5, rowbias=rowbias, rowmap=rowmap)
    ep = dict(res1=x, r1=s_acc, s_acc=s_acc)
    if rowbias is not None:
        ep.update(rowbias=rowbias, rowmap=rowmap)
    if res2 is not None:
        ep.update(res2=res2, r2=r2)
    return _ff(ctx, pk, ln, **ep)


def _ff(ctx: Ctx, pk, x_norm: torch.Tensor, **epilogue) -> torch.Tensor:
    """GEGLU feed-forward: [T,C] -> [T,4C] (fused gelu gate) -> [T,C] with the caller's epilogue"""
    T, Cc = x_norm.shape
    inner = pk.wo.shape[1]
    g = ctx.new(T, inner)
    ops.gemm(x_norm, pk.wp, g, M=T, N=2 * inner, K=Cc, bias=pk.bp, geglu=pk.half)
    out = ctx.new(T, pk.wo.shape[0])
    ops.gemm(g, pk.wo, out, M=T, N=pk.wo.shape[0], K=inner, bias=pk.bo, **epilogue)
    return out


def _cross_literal(block, ctx: Ctx, h1: torch.Tensor, rowmap, first_ctx: int) -> torch.Tensor:
    """attn2 as written (patch/patch.py:526-549, :660-668) for a context of Lk > 1 tokens: norm2 -> to_q; to_k | to_v of the
    context tokens; every row against the Lk keys of the context `rowmap` selects (counted from entry `first_ctx`);
    to_out + residual.  The one-token case never comes here (folded into a row bias by _cross_tables)."""
    pk = block._pk
    if not hasattr(pk, "x2"):
        with _replay.invariant():      # built once from the weights: a constant of any plan being recorded
            pk.x2 = block.attn2.pack_cross(block.norm2)
    x2, (T, Cc), Lk = pk.x2, h1.shape, ctx.cross_Lk
    ln2 = ops.layernorm(h1, None, None, 1e-5)
    q = ctx.new(T, Cc)
    ops.gemm(ln2, x2.wq, q, M=T, N=Cc, K=Cc, bias=x2.bq)
    e = ctx.cross_e[first_ctx * Lk:]
    kv = ctx.new(e.shape[0], 2 * Cc)
    ops.gemm(e, x2.wkv, kv, M=e.shape[0], N=2 * Cc, K=e.shape[1])
    att = ctx.new(T, Cc)
    ops.attn_cross(q, kv[:, :Cc], kv[:, Cc:], att, block.attn2.heads, e.shape[0] // Lk, Lk, rowmap)
    out = ctx.new(T, Cc)
    ops.gemm(att, x2.wo, out, M=T, N=Cc, K=Cc, bias=x2.bo, res1=h1)
    return out


class BasicTransformerBlock(nn.Module):
    """spatial transformer block (witness patch/patch.py:390-580)"""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, cross_attention_dim: int):
        super().__init__()
        self.only_cross_attention = False
        self.norm_type = "layer_norm"
        self.pos_embed = None
        self._chunk_size = None
        self._chunk_dim = 0
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, num_attention_heads, attention_head_dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, num_attention_heads, attention_head_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)
        # patch API state (lkgd_amd/patch.py)
        self.enable_joint_attention = False
        self.joint_scale = 1.0

    def pack(self, model):
        # LayerNorm affines are folded into the Linears they feed: the norm kernels only normalise
        pk = SimpleNamespace(a1=self.attn1.pack_self(self.norm1), ff=self.ff.pack(self.norm3))
        pk.xoff = model._register_cross(self.attn2)
        if hasattr(self, "attn1n"):
            pk.a1n = self.attn1n.pack_self(self.norm1)
            _pack_joint_post(self, pk, spatial=True)
        if hasattr(self, "conv_fuse"):      # FSM hook (lkgd_amd/patch_FSM.py)
            pk.wfuse, pk.bfuse = pack_conv3x3(self.conv_fuse.weight), _f32(self.conv_fuse.bias)
        pk.var = {}                         # masked-LoRA weight variants, built on demand (_attn_variant)
        self._pk = pk

    @_trace.traced("spatial_transformer_block")
    def run(self, ctx: Ctx, h: torch.Tensor) -> torch.Tensor:
        pk, T, Cc = self._pk, h.shape[0], h.shape[1]
        heads = self.attn1.heads
        # norm1 folded into the QKV projection where the row-panel program runs it (K = 320 at many rows): the normalised
        # tokens are never written; the joint branch reads them again and keeps the LayerNorm pass
        joint = self.enable_joint_attention and hasattr(self, "attn1n")
        fold = ctx.lora is None and pk.a1.cs is not None and ops.gemm_ln_ok(T, 3 * Cc, Cc) and not joint
        # LayerNorm + Q|K|V in one launch of the fused-kernel skeleton (qkv_fused.hip: the 72x128 and 36x64 levels)
        one = ctx.lora is None and not joint and ops.ln_qkv_ok(T, 3 * Cc, Cc)
        ln = None if (fold or one) else ops.layernorm(h, None, None, 1e-5)
        qkv = ctx.new(T, 3 * Cc)
        if one:
            if getattr(pk.a1, "wlnqkv", None) is None:
                from .packing import pack_ln_proj
                with _replay.invariant():      # built once from the weights: a constant of any plan being recorded
                    pk.a1.wlnqkv = pack_ln_proj(pk.a1.wqkv, pk.a1.bqkv)
            ops.ln_qkv(h, pk.a1.wlnqkv, qkv)
        elif fold:
            ops.gemm(h, pk.a1.wqkv, qkv, M=T, N=3 * Cc, K=Cc, bias=pk.a1.bqkv, ln=(pk.a1.cs, 1e-5))
        elif ctx.lora is None:
            ops.gemm(ln, pk.a1.wqkv, qkv, M=T, N=3 * Cc, K=Cc, bias=pk.a1.bqkv)
        else:
            va = [_attn_variant(self, "a1", ctx, i, True) for i in range(len(ctx.lora.runs))]
            _gemm_runs(ctx, ln, qkv, lambda i: (va[i].wqkv, va[i].bqkv, None), N=3 * Cc, K=Cc)
        # flip=True joint attention on a frame-sharded rank: attn1n's K | V rows come from the mirror shard.  Its projection runs FIRST
        # and the exchange is issued here, so that it travels under this branch's attention and out-projection (the one place of the
        # sharded forward where an exchange's consumer is not the very next op)
        pre = self._joint_start(ctx, ln) if (joint and ctx.flip_mirror) else None
        att = ctx.new(T, Cc)
        ops.attn_spatial(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], att, ctx.N, ctx.HW, heads)
        h1 = ctx.new(T, Cc)
        if ctx.lora is not None:
            if getattr(self, "_lkgd_fsm", False) and self.enable_joint_attention:
                raise LkgdHipError("masked LoRA together with the FSM hook is not supported")
            _gemm_runs(ctx, att, h1, lambda i: (va[i].wo, va[i].bo, ctx.xb_runs[i][ctx.b0:, pk.xoff:pk.xoff + Cc]),
                       N=Cc, K=Cc, rowmap=ops.rowmap_div(ctx.F * ctx.HW), res1=h)
            if self.enable_joint_attention and hasattr(self, "attn1n"):
                h1 = self._joint(ctx, ln, h1, pre)
            return self._tail(ctx, h1)
        if getattr(self, "_lkgd_fsm", False) and self.enable_joint_attention:
            # the track fuse reads attn1(x) + x BEFORE cross-attention: the folded attn2 bias is added by its last kernels
            ops.gemm(att, pk.a1.wo, h1, M=T, N=Cc, K=Cc, bias=pk.a1.bo, res1=h)
            return self._tail(ctx, self._fsm(ctx, h1))
        # attn1 out-projection + residual + (attn2 == per-batch bias, norm2/Q/K are dead for one key token)
        ops.gemm(att, pk.a1.wo, h1, M=T, N=Cc, K=Cc, bias=pk.a1.bo, res1=h,
                 rowbias=ctx.xb_all[ctx.b0:, pk.xoff:pk.xoff + Cc], rowmap=ops.rowmap_div(ctx.F * ctx.HW))
        if self.enable_joint_attention and hasattr(self, "attn1n"):
            h1 = self._joint(ctx, ln, h1, pre)
        return self._tail(ctx, h1)

    def _tail(self, ctx: Ctx, h1: torch.Tensor) -> torch.Tensor:
        """(literal attn2 for a multi-token context,) norm3 + feed-forward + residual"""
        if ctx.cross_Lk > 1:
            h1 = _cross_literal(self, ctx, h1, ops.rowmap_div(ctx.F * ctx.HW), ctx.b0)
        return _ff_ln(ctx, self._pk.ff, h1)

    def _joint_start(self, ctx: Ctx, ln: torch.Tensor):
        """Q | K | V of attn1n (per-entry LoRA variants where the model carries them); on a flip=True frame-sharded rank also the
        start of the K | V exchange with the mirror shard.  Returns (qkv, vj, k, v, finish)"""
        pk, T, Cc = self._pk, ln.shape[0], ln.shape[1]
        if ctx.spatial_partner is None:
            raise LkgdHipError("joint attention enabled but no joint_attn_mask set (patch.set_joint_attention_mask)")
        qkv = ctx.new(T, 3 * Cc)
        vj = None
        if ctx.lora is None:
            ops.gemm(ln, pk.a1n.wqkv, qkv, M=T, N=3 * Cc, K=Cc, bias=pk.a1n.bqkv)
        else:
            vj = [_attn_variant(self, "a1n", ctx, i, True) for i in range(len(ctx.lora.runs))]
            _gemm_runs(ctx, ln, qkv, lambda i: (vj[i].wqkv, vj[i].bqkv, None), N=3 * Cc, K=Cc)
        kk, vv, finish = qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], None
        if ctx.flip_mirror:
            # the partner's frame F-1-f lives on the mirror shard: trade the K | V rows of this block with it (one all-to-all whose
            # only block is the mirror's).  The projections - also the partner-side LoRA variants - were computed where the rows live
            from . import dist as _dist
            kv_own = ctx.new(T, 2 * Cc)
            src = qkv[:, Cc:]
            _dist._step(lambda: kv_own.copy_(src))
            kv_mirror, finish = ctx.shard.mirror_start(kv_own)
            kk, vv = kv_mirror[:, :Cc], kv_mirror[:, Cc:]
        return qkv, vj, kk, vv, finish

    def _joint(self, ctx: Ctx, ln: torch.Tensor, h1: torch.Tensor, pre=None) -> torch.Tensor:
        """joint attention attn1n with the partner batch entry's K/V (patch/patch.py:438-501); the post step
        (conv1n / scale1n / conv_fuse) is folded into the out-projection, so the branch ends in ONE GEMM epilogue
        h1 + joint_scale * post(attn1n(...)).  ``pre``: what _joint_start returned when run() called it ahead of the main branch"""
        pk, T, Cc = self._pk, ln.shape[0], ln.shape[1]
        qkv, vj, kk, vv, finish = pre if pre is not None else self._joint_start(ctx, ln)
        if finish is not None:
            finish()                   # the mirror shard's K | V have arrived (the launch stream waits for the exchange here)
        att = ctx.new(T, Cc)
        ops.attn_spatial(qkv[:, :Cc], kk, vv, att, ctx.N, ctx.HW, self.attn1n.heads, kv_batch_map=ctx.spatial_partner)
        out = ctx.new(T, Cc)
        js = float(self.joint_scale)
        if pk.post != "conv_fuse":
            if vj is None:
                ops.gemm(att, pk.jw, out, M=T, N=Cc, K=Cc, bias=pk.jb, s_acc=js, res1=h1)
            else:
                _gemm_runs(ctx, att, out, lambda i: (vj[i].jw, vj[i].jb, None), N=Cc, K=Cc, s_acc=js, res1=h1)
            return out
        if vj is not None:
            raise LkgdHipError("masked LoRA with post='conv_fuse' joint layers is not supported")
        # conv_fuse: the i-th masked and i-th unmasked entry blocks are fused pairwise (:488-493) - one two-source GEMM
        # per entry block, own rows | partner rows along K
        if ctx.joint_blocks is None:
            raise LkgdHipError("post='conv_fuse' needs as many masked as unmasked entries in joint_attn_mask")
        for e0, n, p0, masked in ctx.joint_blocks:
            r0, r1, q0 = e0 * ctx.HW, (e0 + n) * ctx.HW, p0 * ctx.HW
            ops.gemm(att[r0:r1], pk.jw_m if masked else pk.jw_u, out[r0:r1], M=r1 - r0, N=Cc, K=2 * Cc,
                     a1=att[q0:q0 + (r1 - r0)], csplit=Cc, bias=pk.jb_m if masked else pk.jb_u, s_acc=js,
                     res1=h1[r0:r1])
        return out

    def _fsm(self, ctx: Ctx, hA: torch.Tensor) -> torch.Tensor:
        """track-guided fuse between even (src) and odd (dst) batch entries (patch/patch_FSM.py:380-441): gather dst
        tokens at the tracked points, scatter-mean onto the src grid, 3x3 ``conv_fuse`` over cat(src, reduced), scatter
        the fused partner half back along the tracks; returns hA + fused + cross-attention bias."""
        pk, T, Cc = self._pk, hA.shape[0], hA.shape[1]
        if not hasattr(pk, "wfuse"):
            raise LkgdHipError("FSM hook enabled but conv_fuse is missing (patch_FSM.initialize_joint_layers)")
        if ctx.N % 2:
            raise LkgdHipError("FSM hook: batch*frames must be even (hidden_states[::2] / [1::2] pairs)")
        if ctx.shard is not None:
            # the hook pairs rows 2k / 2k+1 of the flattened (batch, frame) axis of the WHOLE call.  A rank's local pairs are those
            # pairs when every local entry starts on an even global row: frame slices cut at even frames (dist.make_plan(...,
            # frame_unit=2), which DistDenoiser selects when the model carries the hook) of clips with an even frame count
            if any(((ctx.b0 + b) * ctx.F_total + ctx.f0 - b * ctx.F) % 2 for b in range(ctx.B)):
                raise LkgdHipError(f"FSM hook under sharding: the rank's frames [{ctx.f0}, {ctx.f0 + ctx.F}) of {ctx.F_total} split a "
                                   "(2k, 2k+1) frame pair - build the shard plan with frame_unit=2 and an even frame count")
        from .patch_FSM import track_tables
        with _replay.invariant():            # set once per clip (update_patch): constants of the clip's plan
            fwd, bwd = track_tables(self, ctx)
        pairs, HW = ctx.N // 2, ctx.HW
        srcf, red = ctx.new(pairs * HW, Cc), ctx.new(pairs * HW, Cc)
        ops.fsm_rows(hA, srcf, pairs=pairs, HW=HW, C_=Cc, a_rows=(2 * HW, 0), o_rows=(HW, 0))
        ops.fsm_rows(hA, red, pairs=pairs, HW=HW, C_=Cc, a_rows=(2 * HW, HW), o_rows=(HW, 0), csr=fwd)
        fused = ctx.new(pairs * HW, 2 * Cc)
        ops.gemm(srcf, pk.wfuse, fused, M=pairs * HW, N=2 * Cc, K=18 * Cc, a1=red, csplit=Cc, bias=pk.bfuse,
                 mode=ops.A_CONV3X3, Cin=2 * Cc, conv=(ctx.H, ctx.W, ctx.H, ctx.W, 1, 0))
        out = ctx.new(T, Cc)
        xb = ctx.xb_all[ctx.b0:, pk.xoff:pk.xoff + Cc]
        ops.fsm_rows(fused[:, :Cc], out, pairs=pairs, HW=HW, C_=Cc, a_rows=(HW, 0), o_rows=(2 * HW, 0), res=hA,
                     r_rows=(2 * HW, 0), bias=xb, bias_map=(2, 0, ctx.F))
        ops.fsm_rows(fused[:, Cc:], out, pairs=pairs, HW=HW, C_=Cc, a_rows=(HW, 0), o_rows=(2 * HW, HW), res=hA,
                     r_rows=(2 * HW, HW), bias=xb, bias_map=(2, 1, ctx.F), csr=bwd)
        return out


class TemporalBasicTransformerBlock(nn.Module):
    """temporal transformer block (witness patch/patch.py:582-686); rows stay (b, f, s) - no regroup copies"""

    def __init__(self, dim: int, time_mix_inner_dim: int, num_attention_heads: int, attention_head_dim: int,
                 cross_attention_dim: int):
        super().__init__()
        if dim != time_mix_inner_dim:
            raise LkgdHipError("TemporalBasicTransformerBlock: dim != time_mix_inner_dim is not used by SVD")
        self.is_res = True
        self._chunk_size = None
        self._chunk_dim = 0
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(time_mix_inner_dim, None, num_attention_heads, attention_head_dim)
        self.norm2 = nn.LayerNorm(time_mix_inner_dim)
        self.attn2 = Attention(time_mix_inner_dim, cross_attention_dim, num_attention_heads, attention_head_dim)
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)
        self.enable_joint_attention = False
        self.joint_scale = 1.0

    def pack(self, model):
        pk = SimpleNamespace(ffin=self.ff_in.pack(self.norm_in), a1=self.attn1.pack_self(self.norm1),
                             ff=self.ff.pack(self.norm3))
        pk.xoff = model._register_cross(self.attn2)
        if hasattr(self, "attn1n"):
            pk.a1n = self.attn1n.pack_self(self.norm1)
            _pack_joint_post(self, pk, spatial=False)
        pk.var = {}
        self._pk = pk

    @_trace.traced("temporal_transformer_block")
    def run(self, ctx: Ctx, h_s: torch.Tensor, posemb: torch.Tensor, alpha: float, order: str) -> torch.Tensor:
        """h_s: output of the spatial block; returns alpha*h_s + (1-alpha)*temporal(h_s + posemb[f])"""
        pk, T, Cc = self._pk, h_s.shape[0], h_s.shape[1]
        fmap = ops.rowmap_div_mod(ctx.HW, ctx.F)
        m1 = _ff_ln(ctx, pk.ffin, h_s, rowbias=posemb, rowmap=fmap)               # ff_in(norm_in(m0)) + m0, m0 = h_s + pos
        att = ctx.new(T, Cc)
        va = None
        att_joint = None       # attn1n's attention output where the sharded paths below compute it (frames on several GPUs)
        fin_att = fin_joint = None     # pending halves of the re-sharding exchanges that bring attention outputs back to frame slices
        # LayerNorm + QKV + attention over the frames in one kernel where it exists (C = 320: the 72x128 level); the joint
        # branch reads the normalised tokens again, masked LoRA runs per-entry weights, a sharded rank gathers frames: unfused
        joint = self.enable_joint_attention and hasattr(self, "attn1n")
        base_ok = ctx.lora is None and not ctx.frames_sharded and ops.tattn_front_ok(Cc, self.attn1.heads, ctx.F, ctx.HW)
        fused = base_ok and not joint
        from . import dist as _dist
        # (a level with fewer pixels than shards - the 1x1 level of the tiny test nets - keeps the gathered form)
        resharded = ctx.frames_sharded and not _dist.TEMPORAL_GATHER and ctx.HW >= ctx.shard.plan.frame_shards
        # ... and the out-projection, its residual and the folded cross-attention table in the same launch (attn_tblock.hip);
        # with the joint branch on, the main branch still runs that way and only the joint branch's input is normalised apart
        one_launch = base_ok and ops.tattn_block_ok(Cc, self.attn1.heads, ctx.F, ctx.HW)
        # (the 36x64 level: LayerNorm + Q|K|V in one launch, then the attention kernel)
        ln_qkv_one = (not (fused or resharded or one_launch) and ctx.lora is None and not ctx.frames_sharded and not joint and
                      ops.ln_qkv_ok(T, 3 * Cc, Cc))
        ln1 = None if (((fused or resharded or one_launch) and not (one_launch and joint)) or ln_qkv_one) else \
            ops.layernorm(m1, None, None, 1e-5)
        if one_launch:
            pass
        elif fused:
            if getattr(pk.a1, "wfront", None) is None:
                from .packing import pack_tfront
                with _replay.invariant():      # built once from the weights: a constant of any plan being recorded
                    pk.a1.wfront = pack_tfront(pk.a1.wqkv, self.attn1.heads)
            ops.tattn_front(m1, pk.a1.wfront, pk.a1.bqkv, att, ctx.B, ctx.F, ctx.HW, self.attn1.heads)
        elif ctx.lora is not None and not ctx.frames_sharded:
            va = [_attn_variant(self, "a1", ctx, i, False) for i in range(len(ctx.lora.runs))]
            qkv = ctx.new(T, 3 * Cc)
            _gemm_runs(ctx, ln1, qkv, lambda i: (va[i].wqkv, va[i].bqkv, None), N=3 * Cc, K=Cc)
            ops.attn_temporal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], att, ctx.B, ctx.F, ctx.HW,
                              self.attn1.heads)
        elif not ctx.frames_sharded:
            qkv = ctx.new(T, 3 * Cc)
            if ln_qkv_one:
                if getattr(pk.a1, "wlnqkv", None) is None:
                    from .packing import pack_ln_proj
                    with _replay.invariant():      # built once from the weights: a constant of any plan being recorded
                        pk.a1.wlnqkv = pack_ln_proj(pk.a1.wqkv, pk.a1.bqkv)
                ops.ln_qkv(m1, pk.a1.wlnqkv, qkv)
            else:
                ops.gemm(ln1, pk.a1.wqkv, qkv, M=T, N=3 * Cc, K=Cc, bias=pk.a1.bqkv)
            ops.attn_temporal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], att, ctx.B, ctx.F, ctx.HW,
                              self.attn1.heads)
        elif resharded:
            # frames of the clip live on several GPUs: re-shard by PIXELS around the attention (all-to-all), so that this rank
            # holds all F frames of its pixel slice - LayerNorm, Q|K|V and the attention then run once per token, fused where
            # the kernel applies - and bring the attention output back to frame slices (lkgd_amd/dist.py)
            m1p = ctx.shard.to_pixels(m1, ctx.HW)              # rows (entry, frame, pixel of this rank's slice)
            Tp, Ft = m1p.shape[0], ctx.F_total
            pxl = Tp // (Ft * ctx.B)
            attp = ctx.new(Tp, Cc)
            ln1p = None
            if ctx.lora is not None:
                # masked LoRA (round 6): a rank holds its slice of every clip of its CFG half, so the per-entry weight variants
                # apply to its entries unchanged - one launch per entry run, rows per entry = all F frames of the pixel slice
                va = [_attn_variant(self, "a1", ctx, i, False) for i in range(len(ctx.lora.runs))]
                ln1p = ops.layernorm(m1p, None, None, 1e-5)
                qkv = ctx.new(Tp, 3 * Cc)
                _gemm_runs(ctx, ln1p, qkv, lambda i: (va[i].wqkv, va[i].bqkv, None), N=3 * Cc, K=Cc, rows=Ft * pxl)
                ops.attn_temporal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], attp, ctx.B, Ft, pxl, self.attn1.heads)
            elif ops.tattn_front_ok(Cc, self.attn1.heads, Ft, pxl):
                if getattr(pk.a1, "wfront", None) is None:
                    from .packing import pack_tfront
                    with _replay.invariant():      # built once from the weights: a constant of any plan being recorded
                        pk.a1.wfront = pack_tfront(pk.a1.wqkv, self.attn1.heads)
                ops.tattn_front(m1p, pk.a1.wfront, pk.a1.bqkv, attp, ctx.B, Ft, pxl, self.attn1.heads)
            else:
                ln1p = ops.layernorm(m1p, None, None, 1e-5)
                qkv = ctx.new(Tp, 3 * Cc)
                ops.gemm(ln1p, pk.a1.wqkv, qkv, M=Tp, N=3 * Cc, K=Cc, bias=pk.a1.bqkv)
                ops.attn_temporal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], attp, ctx.B, Ft, pxl, self.attn1.heads)
            # the attention output goes back to frame slices: issued here, awaited right before the out-projection reads it - with the
            # joint branch on, that branch's projection and attention run beside the transfer (RCCL: an asynchronous all-to-all)
            att, fin_att = ctx.shard.to_frames_start(attp, ctx.HW)
            if joint:
                # the joint branch in the same pixel layout: all frames of the partner ENTRY's pixels are on this rank (a rank
                # holds its slice of every clip of its CFG half), so attn1n needs no exchange beyond bringing its output back
                if ctx.temporal_partner is None:
                    raise LkgdHipError("joint attention enabled but no joint_attn_mask set")
                if ln1p is None:
                    ln1p = ops.layernorm(m1p, None, None, 1e-5)
                qkvj, attjp = ctx.new(Tp, 3 * Cc), ctx.new(Tp, Cc)
                if ctx.lora is None:
                    ops.gemm(ln1p, pk.a1n.wqkv, qkvj, M=Tp, N=3 * Cc, K=Cc, bias=pk.a1n.bqkv)
                else:
                    vjp = [_attn_variant(self, "a1n", ctx, i, False) for i in range(len(ctx.lora.runs))]
                    _gemm_runs(ctx, ln1p, qkvj, lambda i: (vjp[i].wqkv, vjp[i].bqkv, None), N=3 * Cc, K=Cc, rows=Ft * pxl)
                ops.attn_temporal(qkvj[:, :Cc], qkvj[:, Cc:2 * Cc], qkvj[:, 2 * Cc:], attjp, ctx.B, Ft, pxl, self.attn1n.heads,
                                  kv_b_map=ctx.temporal_partner)
                att_joint, fin_joint = ctx.shard.to_frames_start(attjp, ctx.HW)    # ... and this one beside the main out-projection
        else:
            # LKGD_TEMPORAL_GATHER=1: local queries against the keys / values of ALL frames.  The
            # normalised hidden states are gathered (C channels) and K|V projected here for every frame: half the
            # bytes of gathering K|V, for a 2C x C GEMM on F*HW rows
            ln1f = ctx.shard.gather(ln1)
            Tf = ln1f.shape[0]
            q, kvf = ctx.new(T, Cc), ctx.new(Tf, 2 * Cc)
            if ctx.lora is None:
                ops.gemm(ln1, pk.a1.wqkv[:Cc], q, M=T, N=Cc, K=Cc, bias=pk.a1.bqkv[:Cc])
                ops.gemm(ln1f, pk.a1.wqkv[Cc:], kvf, M=Tf, N=2 * Cc, K=Cc, bias=pk.a1.bqkv[Cc:])
            else:
                va = [_attn_variant(self, "a1", ctx, i, False) for i in range(len(ctx.lora.runs))]
                _gemm_runs(ctx, ln1, q, lambda i: (va[i].wqkv[:Cc], va[i].bqkv[:Cc], None), N=Cc, K=Cc)
                _gemm_runs(ctx, ln1f, kvf, lambda i: (va[i].wqkv[Cc:], va[i].bqkv[Cc:], None), N=2 * Cc, K=Cc,
                           rows=ctx.F_total * ctx.HW)
            ops.attn_temporal(q, kvf[:, :Cc], kvf[:, Cc:], att, ctx.B, ctx.F_total, ctx.HW, self.attn1.heads,
                              Fq=ctx.F)
            if joint:         # attn1n: local query frames against the partner entry's keys / values of ALL frames
                if ctx.temporal_partner is None:
                    raise LkgdHipError("joint attention enabled but no joint_attn_mask set")
                qj, kvj, att_joint = ctx.new(T, Cc), ctx.new(Tf, 2 * Cc), ctx.new(T, Cc)
                if ctx.lora is None:
                    ops.gemm(ln1, pk.a1n.wqkv[:Cc], qj, M=T, N=Cc, K=Cc, bias=pk.a1n.bqkv[:Cc])
                    ops.gemm(ln1f, pk.a1n.wqkv[Cc:], kvj, M=Tf, N=2 * Cc, K=Cc, bias=pk.a1n.bqkv[Cc:])
                else:
                    vjg = [_attn_variant(self, "a1n", ctx, i, False) for i in range(len(ctx.lora.runs))]
                    _gemm_runs(ctx, ln1, qj, lambda i: (vjg[i].wqkv[:Cc], vjg[i].bqkv[:Cc], None), N=Cc, K=Cc)
                    _gemm_runs(ctx, ln1f, kvj, lambda i: (vjg[i].wqkv[Cc:], vjg[i].bqkv[Cc:], None), N=2 * Cc, K=Cc,
                               rows=ctx.F_total * ctx.HW)
                ops.attn_temporal(qj, kvj[:, :Cc], kvj[:, Cc:], att_joint, ctx.B, ctx.F_total, ctx.HW, self.attn1n.heads,
                                  kv_b_map=ctx.temporal_partner, Fq=ctx.F)
        xtab = ctx.xb_all[:, pk.xoff:pk.xoff + Cc]
        if isinstance(order, tuple):
            xmap = order                      # explicit context-row map (tests drive single blocks this way)
        elif order == "interleaved_0_27":
            # global row i = (b0 + b)*HW + s of the [B_total*HW, F, C] regroup uses context i % B_total
            xmap = (ctx.F * ctx.HW, ctx.HW, ctx.HW, ctx.B_total, (ctx.b0 * ctx.HW) % ctx.B_total)
        elif order == "batch_major":
            xmap, xtab = ops.rowmap_div(ctx.F * ctx.HW), ctx.xb_all[ctx.b0:, pk.xoff:pk.xoff + Cc]
        else:
            raise ValueError(order)
        m2 = ctx.new(T, Cc)
        if fin_att is not None:
            fin_att()
        if one_launch:
            if getattr(pk.a1, "wblock", None) is None:
                from .packing import pack_tblock
                with _replay.invariant():      # built once from the weights: a constant of any plan being recorded
                    pk.a1.wblock = pack_tblock(pk.a1.wqkv, pk.a1.bqkv, pk.a1.wo, self.attn1.heads)
            ops.tattn_block(m1, pk.a1.wblock, pk.a1.bo, m2, ctx.B, ctx.F, ctx.HW, rowbias=xtab, rowmap=xmap)
        elif va is None:
            ops.gemm(att, pk.a1.wo, m2, M=T, N=Cc, K=Cc, bias=pk.a1.bo, res1=m1, rowbias=xtab, rowmap=xmap)
        else:
            b_off = ctx.b0 if order == "batch_major" else 0
            _gemm_runs(ctx, att, m2, lambda i: (va[i].wo, va[i].bo, ctx.xb_runs[i][b_off:, pk.xoff:pk.xoff + Cc]),
                       N=Cc, K=Cc, rowmap=xmap, res1=m1)
        if fin_joint is not None:
            fin_joint()
        if self.enable_joint_attention and hasattr(self, "attn1n"):
            m2 = self._joint(ctx, ln1, m2, att_joint)
        if ctx.cross_Lk > 1:                  # literal attn2 over the time context (same row -> context map as the folded bias)
            m2 = _cross_literal(self, ctx, m2, xmap, ctx.b0 if order == "batch_major" else 0)
        # ff(norm3(m2)) + m2, then AlphaBlender with the spatial branch - one epilogue
        return _ff_ln(ctx, pk.ff, m2, s_acc=1.0 - alpha, res2=h_s, r2=alpha)

    def _joint(self, ctx: Ctx, ln1: Optional[torch.Tensor], m2: torch.Tensor, att: Optional[torch.Tensor] = None) -> torch.Tensor:
        """temporal joint branch (patch/patch.py:616-658): attn1n over the partner batch entry's frames; ``att``: the attention
        output where the caller computed it already (frame-sharded ranks: over all frames, run())"""
        pk, T, Cc = self._pk, m2.shape[0], m2.shape[1]
        if ctx.temporal_partner is None:
            raise LkgdHipError("joint attention enabled but no joint_attn_mask set")
        vj = None
        if ctx.lora is not None:
            vj = [_attn_variant(self, "a1n", ctx, i, False) for i in range(len(ctx.lora.runs))]
        if att is None:
            qkv = ctx.new(T, 3 * Cc)
            if vj is None:
                ops.gemm(ln1, pk.a1n.wqkv, qkv, M=T, N=3 * Cc, K=Cc, bias=pk.a1n.bqkv)
            else:
                _gemm_runs(ctx, ln1, qkv, lambda i: (vj[i].wqkv, vj[i].bqkv, None), N=3 * Cc, K=Cc)
            att = ctx.new(T, Cc)
            ops.attn_temporal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], att, ctx.B, ctx.F, ctx.HW,
                              self.attn1n.heads, kv_b_map=ctx.temporal_partner)
        out = ctx.new(T, Cc)
        if vj is None:   # post folded in; joint_scale is not applied here
            ops.gemm(att, pk.jw, out, M=T, N=Cc, K=Cc, bias=pk.jb, res1=m2)
        else:
            _gemm_runs(ctx, att, out, lambda i: (vj[i].jw, vj[i].jb, None), N=Cc, K=Cc, res1=m2)
        return out


class AlphaBlender(nn.Module):
    def __init__(self, alpha: float = 0.5):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([alpha], dtype=torch.float32))


class TransformerSpatioTemporalModel(nn.Module):
    def __init__(self, num_attention_heads: int, attention_head_dim: int, in_channels: int, num_layers: int = 1,
                 cross_attention_dim: int = 1024):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        if inner != in_channels:
            raise LkgdHipError("TransformerSpatioTemporalModel: inner_dim != in_channels is not used by SVD")
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim)
            for _ in range(num_layers)])
        self.temporal_transformer_blocks = nn.ModuleList([
            TemporalBasicTransformerBlock(inner, inner, num_attention_heads, attention_head_dim, cross_attention_dim)
            for _ in range(num_layers)])
        self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_mixer = AlphaBlender(0.5)
        self.proj_out = nn.Linear(inner, in_channels)
        self.time_context_order = None
        self._posemb = {}

    def pack(self, model):
        self._pk = SimpleNamespace(gn=(_f32(self.norm.weight), _f32(self.norm.bias)),
                                   win=pack_linear(self.proj_in.weight), bin=_f32(self.proj_in.bias),
                                   wout=pack_linear(self.proj_out.weight), bout=_f32(self.proj_out.bias))
        self.time_pos_embed.pack()
        self._posemb = {}
        self._alpha = None
        for b in self.transformer_blocks:
            b.pack(model)
        for b in self.temporal_transformer_blocks:
            b.pack(model)

    def _pos(self, ctx: Ctx) -> torch.Tensor:
        """frame positional embedding table [F, C]: step- and input-invariant -> computed once per frame count"""
        e = self._posemb.get(ctx.F_total)
        if e is None:
            t = torch.arange(ctx.F_total, dtype=torch.float32, device=ctx.device)
            e = self.time_pos_embed.run(ops.timestep_embedding(t, self.in_channels))
            self._posemb[ctx.F_total] = e
        return e

    @_trace.traced("transformer")
    def run(self, ctx: Ctx, x: torch.Tensor) -> torch.Tensor:
        pk, T, Cc = self._pk, x.shape[0], x.shape[1]
        if self._alpha is None:
            with _replay.invariant():      # built once from the weights: a constant of any plan being recorded
                self._alpha = float(torch.sigmoid(self.time_mixer.mix_factor.detach().float()).item())
        n = ops.groupnorm_silu(x, None, ctx.N, ctx.HW, *pk.gn, 1e-6, silu=False)
        h = ctx.new(T, Cc)
        ops.gemm(n, pk.win, h, M=T, N=Cc, K=Cc, bias=pk.bin)
        posemb = self._pos(ctx)[ctx.f0:ctx.f0 + ctx.F]
        order = self.time_context_order or TIME_CONTEXT_ORDER
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            h_s = blk.run(ctx, h)
            h = tblk.run(ctx, h_s, posemb, self._alpha, order)
        out = ctx.new(T, Cc)
        ops.gemm(h, pk.wout, out, M=T, N=Cc, K=Cc, bias=pk.bout, res1=x, colstats=ctx.HW)   # -> the next resnet's norm1
        return out


# ----------------------------------------------------------------------------------------------- res blocks
class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, eps: float):
        super().__init__()
        self.eps = eps
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None


class TemporalResnetBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, eps: float):
        super().__init__()
        self.eps = eps
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv3d(in_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, eps: float):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, temb_channels, eps)
        self.temporal_res_block = TemporalResnetBlock(out_channels, out_channels, temb_channels, eps)
        self.time_mixer = AlphaBlender(0.5)

    def pack(self, model):
        s, t = self.spatial_res_block, self.temporal_res_block
        pk = SimpleNamespace(
            n1=(_f32(s.norm1.weight), _f32(s.norm1.bias)), n2=(_f32(s.norm2.weight), _f32(s.norm2.bias)),
            w1=pack_conv3x3(s.conv1.weight.detach()), b1=_f32(s.conv1.bias),
            w2=pack_conv3x3(s.conv2.weight.detach()), b2=_f32(s.conv2.bias),
            tn1=(_f32(t.norm1.weight), _f32(t.norm1.bias)), tn2=(_f32(t.norm2.weight), _f32(t.norm2.bias)),
            tw1=pack_tconv3(t.conv1.weight.detach()), tb1=_f32(t.conv1.bias),
            tw2=pack_tconv3(t.conv2.weight.detach()), tb2=_f32(t.conv2.bias),
            eps=s.eps, teps=t.eps)
        if s.conv_shortcut is not None:
            pk.ws, pk.bs = pack_linear(s.conv_shortcut.weight.detach()), _f32(s.conv_shortcut.bias)
        pk.toff_s = model._register_temb(s.time_emb_proj)
        pk.toff_t = model._register_temb(t.time_emb_proj)
        pk.alpha = None
        self._pk = pk

    @_trace.traced("resblock")
    def run(self, ctx: Ctx, x0: torch.Tensor, x1: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x = cat(x0, x1) on channels (skip connection folded into the consumers' gathers)"""
        pk = self._pk
        if pk.alpha is None:
            with _replay.invariant():      # built once from the weights: a constant of any plan being recorded
                pk.alpha = float(torch.sigmoid(self.time_mixer.mix_factor.detach().float()).item())
        T, Cin, Cout = ctx.T, self.in_channels, self.out_channels
        c0 = x0.shape[1]
        bmap = ops.rowmap_div(ctx.F * ctx.HW)
        geo = (ctx.H, ctx.W, ctx.H, ctx.W, 1, 0)
        # --- spatial ResnetBlock2D
        n1 = ops.groupnorm_silu(x0, x1, ctx.N, ctx.HW, *pk.n1, pk.eps)
        h = ctx.new(T, Cout)
        ops.gemm(n1, pk.w1, h, M=T, N=Cout, K=9 * Cin, bias=pk.b1, mode=ops.A_CONV3X3, Cin=Cin, conv=geo,
                 rowbias=ctx.temb_all[:, pk.toff_s:pk.toff_s + Cout], rowmap=bmap)
        n2 = ops.groupnorm_silu(h, None, ctx.N, ctx.HW, *pk.n2, pk.eps)
        if hasattr(pk, "ws"):
            sc = ctx.new(T, Cout)
            ops.gemm(x0, pk.ws, sc, M=T, N=Cout, K=Cin, a1=x1, csplit=c0, bias=pk.bs)
        else:
            sc = x0
        s = ctx.new(T, Cout)
        ops.gemm(n2, pk.w2, s, M=T, N=Cout, K=9 * Cout, bias=pk.b2, mode=ops.A_CONV3X3, Cin=Cout, conv=geo, res1=sc)
        # --- TemporalResnetBlock on [B, C, F, H, W] == the same tokens; GroupNorm statistics span all F frames
        n3 = self._temporal_norm(ctx, s, pk.tn1, pk.teps)
        # sharded: n3 / n4 are [F + 2] frame buffers (own frames + one halo frame per side), output frames start at slot 1
        tgeo = (ctx.F + 2, ctx.HW, ctx.F, 1) if ctx.frames_sharded else (ctx.F, ctx.HW, ctx.F, 0)
        h = ctx.new(T, Cout)
        ops.gemm(n3, pk.tw1, h, M=T, N=Cout, K=3 * Cout, bias=pk.tb1, mode=ops.A_TCONV3, Cin=Cout,
                 tconv=tgeo, rowbias=ctx.temb_all[:, pk.toff_t:pk.toff_t + Cout], rowmap=bmap, colstats=ctx.F * ctx.HW)   # -> tn2
        n4 = self._temporal_norm(ctx, h, pk.tn2, pk.teps)
        out = ctx.new(T, Cout)
        # temporal = s + conv2(..); AlphaBlender: alpha*s + (1-alpha)*temporal = s + (1-alpha)*conv2(..)
        ops.gemm(n4, pk.tw2, out, M=T, N=Cout, K=3 * Cout, bias=pk.tb2, mode=ops.A_TCONV3, Cin=Cout,
                 tconv=tgeo, s_acc=1.0 - pk.alpha, res1=s, colstats=ctx.HW)    # -> the next block's (spatial) GroupNorm
        return out

    @staticmethod
    def _temporal_norm(ctx: Ctx, x: torch.Tensor, affine, eps: float) -> torch.Tensor:
        """GroupNorm over (C/32, F, H, W) + SiLU.  Sharded: all-reduce the [32,2] partial sums, normalise the local frames
        into the middle of an [F + 2]-frame buffer and fetch the neighbours' boundary frames into its two end slots (the
        Conv3d (3,1,1) that follows needs +-1 frame; at the clip's ends the slot stays zero = the conv's padding)."""
        if not ctx.frames_sharded:
            return ops.groupnorm_silu(x, None, ctx.B, ctx.F * ctx.HW, *affine, eps)
        from . import dist as _dist
        # every batch entry gets its own [F + 2]-frame block (the Conv3d's gather addresses entry b at b * (F + 2) * HW);
        # only the clip's two end slots are padding, every other halo slot is overwritten by the exchange
        HW, Cc = ctx.HW, x.shape[1]
        blk, rows = (ctx.F + 2) * HW, ctx.F * HW
        count = float(ctx.F_total) * HW * (Cc // 32)
        buf = torch.empty(ctx.B * blk, Cc, dtype=torch.float16, device=x.device)
        plan = ctx.shard.plan
        k, si = plan.frame_shards, plan.shard_index
        if _dist.GN_HALO_FUSED and _dist._HALO_ALLGATHER:
            # ONE collective per temporal GroupNorm + Conv3d: the raw boundary frames travel with the partial sums, every rank
            # adds the sums in rank order and normalises the halo frames it received itself (lkgd_amd/dist.py)
            sums = ops.groupnorm_sums(x, None, ctx.B, rows)
            got = ctx.shard.halo_raw([x[b * rows:b * rows + HW] for b in range(ctx.B)],
                                     [x[(b + 1) * rows - HW:(b + 1) * rows] for b in range(ctx.B)], sums)
            n, W = HW * Cc, 2 * HW * Cc + _dist.SUMS_SLOT
            parts = got.view(-1)[2 * n:].view(torch.float32)              # rank 0 / entry 0's sums; strides in floats below
            stats = ops.groupnorm_finalize_parts(parts, k, ctx.B * W // 2, ctx.B, W // 2, count, eps)
            segs, pads = [], []   # one launch: own frames of every entry + the neighbours' raw boundary frames
            for b in range(ctx.B):
                segs.append((x[b * rows:(b + 1) * rows], buf[b * blk + HW:(b + 1) * blk - HW], b))
                if si == 0:
                    pads.append(buf[b * blk:b * blk + HW])
                else:             # the previous shard's LAST frame
                    segs.append((got[si - 1, b, n:2 * n].view(HW, Cc), buf[b * blk:b * blk + HW], b))
                if si == k - 1:
                    pads.append(buf[(b + 1) * blk - HW:(b + 1) * blk])
                else:             # the next shard's FIRST frame
                    segs.append((got[si + 1, b, :n].view(HW, Cc), buf[(b + 1) * blk - HW:(b + 1) * blk], b))
            if pads:              # the clip's end slots = the Conv3d's zero padding; a replayed step: buf is scratch of the plan
                _dist._step(lambda: [p.zero_() for p in pads])
            ops.groupnorm_apply_segments(segs, stats, *affine, True)
            return buf
        sums = ctx.shard.allreduce(ops.groupnorm_sums(x, None, ctx.B, rows))
        stats = ops.groupnorm_finalize(sums, count, eps)
        pads = []
        for b in range(ctx.B):
            if plan.shard_index == 0:
                pads.append(buf[b * blk:b * blk + ctx.HW])
            if plan.shard_index == plan.frame_shards - 1:
                pads.append(buf[(b + 1) * blk - ctx.HW:(b + 1) * blk])
            ops.groupnorm_apply(x[b * rows:(b + 1) * rows], None, 1, rows, stats[b:b + 1], *affine, True,
                                buf[b * blk + ctx.HW:(b + 1) * blk - ctx.HW])
        if pads:
            _dist._step(lambda: [p.zero_() for p in pads])
        return ctx.shard.halo(buf)


class Downsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.channels = channels
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def pack(self, model):
        self._pk = SimpleNamespace(w=pack_conv3x3(self.conv.weight.detach()), b=_f32(self.conv.bias))

    def run(self, ctx: Ctx, x: torch.Tensor) -> torch.Tensor:
        C_ = self.channels
        Ho, Wo = (ctx.H - 1) // 2 + 1, (ctx.W - 1) // 2 + 1
        out = ctx.new(ctx.N * Ho * Wo, C_)
        ops.gemm(x, self._pk.w, out, M=ctx.N * Ho * Wo, N=C_, K=9 * C_, bias=self._pk.b, mode=ops.A_CONV3X3, Cin=C_,
                 conv=(Ho, Wo, ctx.H, ctx.W, 2, 0))
        ctx.H, ctx.W = Ho, Wo
        return out


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.channels = channels
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def pack(self, model):
        self._pk = SimpleNamespace(w=pack_conv3x3(self.conv.weight.detach()), b=_f32(self.conv.bias))

    def run(self, ctx: Ctx, x: torch.Tensor) -> torch.Tensor:
        C_ = self.channels
        Ho, Wo = ctx.H * 2, ctx.W * 2
        out = ctx.new(ctx.N * Ho * Wo, C_)
        ops.gemm(x, self._pk.w, out, M=ctx.N * Ho * Wo, N=C_, K=9 * C_, bias=self._pk.b, mode=ops.A_CONV3X3, Cin=C_,
                 conv=(Ho, Wo, ctx.H, ctx.W, 1, 1))   # nearest-2x folded into the gather
        ctx.H, ctx.W = Ho, Wo
        return out


# ----------------------------------------------------------------------------------------------- block containers
class _BlockBase(nn.Module):
    has_cross_attention = False

    def pack(self, model):
        for m in self.children():
            for sub in (m if isinstance(m, nn.ModuleList) else [m]):
                sub.pack(model)


class CrossAttnDownBlockSpatioTemporal(_BlockBase):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, transformer_layers_per_block,
                 num_attention_heads, cross_attention_dim, add_downsample):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels, temb_channels, eps)
            for i in range(num_layers)])
        self.attentions = nn.ModuleList([
            TransformerSpatioTemporalModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                                           transformer_layers_per_block, cross_attention_dim)
            for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    @_trace.traced("block")
    def run(self, ctx, h):
        outs = []
        for r, a in zip(self.resnets, self.attentions):
            h = a.run(ctx, r.run(ctx, h))
            outs.append((h, ctx.H, ctx.W))
        if self.downsamplers is not None:
            for d in self.downsamplers:
                h = d.run(ctx, h)
            outs.append((h, ctx.H, ctx.W))
        return h, outs


class DownBlockSpatioTemporal(_BlockBase):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, add_downsample):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels, temb_channels, eps)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    @_trace.traced("block")
    def run(self, ctx, h):
        outs = []
        for r in self.resnets:
            h = r.run(ctx, h)
            outs.append((h, ctx.H, ctx.W))
        if self.downsamplers is not None:
            for d in self.downsamplers:
                h = d.run(ctx, h)
            outs.append((h, ctx.H, ctx.W))
        return h, outs


class UNetMidBlockSpatioTemporal(_BlockBase):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, num_layers=1, transformer_layers_per_block=1,
                 num_attention_heads=1, cross_attention_dim=1280):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps)
                                      for _ in range(num_layers + 1)])
        self.attentions = nn.ModuleList([
            TransformerSpatioTemporalModel(num_attention_heads, in_channels // num_attention_heads, in_channels,
                                           transformer_layers_per_block, cross_attention_dim)
            for _ in range(num_layers)])

    @_trace.traced("block")
    def run(self, ctx, h):
        h = self.resnets[0].run(ctx, h)
        for a, r in zip(self.attentions, self.resnets[1:]):
            h = r.run(ctx, a.run(ctx, h))
        return h


def _up_in(i, num_layers, in_channels, prev_output_channel, out_channels):
    res_skip = in_channels if i == num_layers - 1 else out_channels
    res_in = prev_output_channel if i == 0 else out_channels
    return res_in + res_skip


class UpBlockSpatioTemporal(_BlockBase):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(_up_in(i, num_layers, in_channels, prev_output_channel, out_channels),
                                   out_channels, temb_channels, eps) for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    @_trace.traced("block")
    def run(self, ctx, h, skips: List[torch.Tensor]):
        for r in self.resnets:
            h = r.run(ctx, h, skips.pop())
        if self.upsamplers is not None:
            for u in self.upsamplers:
                h = u.run(ctx, h)
        return h


class CrossAttnUpBlockSpatioTemporal(_BlockBase):
    has_cross_attention = True

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                 transformer_layers_per_block, num_attention_heads, cross_attention_dim, add_upsample):
        super().__init__()
        eps = RESNET_EPS[type(self).__name__]
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(_up_in(i, num_layers, in_channels, prev_output_channel, out_channels),
                                   out_channels, temb_channels, eps) for i in range(num_layers)])
        self.attentions = nn.ModuleList([
            TransformerSpatioTemporalModel(num_attention_heads, out_channels // num_attention_heads, out_channels,
                                           transformer_layers_per_block, cross_attention_dim)
            for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    @_trace.traced("block")
    def run(self, ctx, h, skips: List[torch.Tensor]):
        for r, a in zip(self.resnets, self.attentions):
            h = a.run(ctx, r.run(ctx, h, skips.pop()))
        if self.upsamplers is not None:
            for u in self.upsamplers:
                h = u.run(ctx, h)
        return h


def get_down_block(t, num_layers, in_channels, out_channels, temb_channels, add_downsample, num_attention_heads,
                   cross_attention_dim, transformer_layers_per_block):
    if t == "DownBlockSpatioTemporal":
        return DownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers, add_downsample)
    if t == "CrossAttnDownBlockSpatioTemporal":
        return CrossAttnDownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers,
                                                transformer_layers_per_block, num_attention_heads,
                                                cross_attention_dim, add_downsample)
    raise ValueError(t)


def get_up_block(t, num_layers, in_channels, out_channels, prev_output_channel, temb_channels, add_upsample,
                 num_attention_heads, cross_attention_dim, transformer_layers_per_block):
    if t == "UpBlockSpatioTemporal":
        return UpBlockSpatioTemporal(in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                                     add_upsample)
    if t == "CrossAttnUpBlockSpatioTemporal":
        return CrossAttnUpBlockSpatioTemporal(in_channels, prev_output_channel, out_channels, temb_channels,
                                              num_layers, transformer_layers_per_block, num_attention_heads,
                                              cross_attention_dim, add_upsample)
    raise ValueError(t)


# ----------------------------------------------------------------------------------------------- top level
class _UNetBase(nn.Module):
    _encoder_only = False      # lkgd_amd.controlnet: conv_in, embeddings, down blocks and mid block only

    def __init__(self, config: Optional[UNetConfig] = None, **kw):
        super().__init__()
        cfg = config if config is not None else UNetConfig(**kw)
        self.config = SimpleNamespace(**cfg.__dict__)
        self.sample_size = cfg.sample_size
        boc = tuple(cfg.block_out_channels)
        n = len(cfg.down_block_types)
        if len(cfg.up_block_types) != n or len(boc) != n:
            raise ValueError("Must provide the same number of down/up block types and block_out_channels")
        heads = cfg.num_attention_heads
        heads = (heads,) * n if isinstance(heads, int) else tuple(heads)
        if len(heads) != n:
            raise ValueError("Must provide the same number of `num_attention_heads` as `down_block_types`")
        cross = cfg.cross_attention_dim
        cross = (cross,) * n if isinstance(cross, int) else tuple(cross)
        lpb = [cfg.layers_per_block] * n if isinstance(cfg.layers_per_block, int) else list(cfg.layers_per_block)
        tl = cfg.transformer_layers_per_block
        tlpb = [tl] * n if isinstance(tl, int) else list(tl)
        if cfg.in_channels != 8:
            raise LkgdHipError("conv_in kernel path (LKGD_A_CONV3X3_C8) needs in_channels == 8")

        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        ted = boc[0] * 4
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, ted)
        self.down_blocks = nn.ModuleList()
        if not self._encoder_only:
            self.up_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, t in enumerate(cfg.down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            self.down_blocks.append(get_down_block(t, lpb[i], in_ch, out_ch, ted, i != n - 1, heads[i], cross[i],
                                                   tlpb[i]))
        self._init_extra(cfg)
        self.mid_block = UNetMidBlockSpatioTemporal(boc[-1], ted, transformer_layers_per_block=tlpb[-1],
                                                    num_attention_heads=heads[-1], cross_attention_dim=cross[-1])
        self._pk = None
        self._temb_reg: List[nn.Linear] = []
        self._cross_reg: List[Attention] = []
        if self._encoder_only:
            return
        rboc, rheads = list(reversed(boc)), list(reversed(heads))
        rlpb, rcross, rtlpb = list(reversed(lpb)), list(reversed(cross)), list(reversed(tlpb))
        out_ch = rboc[0]
        self.num_upsamplers = 0
        for i, t in enumerate(cfg.up_block_types):
            prev, out_ch = out_ch, rboc[i]
            in_ch = rboc[min(i + 1, n - 1)]
            self.up_blocks.append(get_up_block(t, rlpb[i] + 1, in_ch, out_ch, prev, ted, i != n - 1, rheads[i],
                                               rcross[i], rtlpb[i]))
            self.num_upsamplers += int(i != n - 1)
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def _init_extra(self, cfg):
        pass

    def _pack_extra(self, pk):
        pass

    # ---- reference API surface -----------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: Optional[str] = None, torch_dtype=None,
                        variant: Optional[str] = None, **_ignored):
        """``ModelMixin.from_pretrained`` [EXT] for a local diffusers directory (``unet/config.json`` + safetensors), as
        /root/reference/utils/util.py:607 / run_inference_svd.py:164 call it; hub / cache / device-map keywords are
        accepted and ignored"""
        from .loading import build_from_pretrained
        return build_from_pretrained(cls, UNetConfig, pretrained_model_name_or_path, subfolder, torch_dtype, variant)

    def save_pretrained(self, save_directory: str, variant: Optional[str] = None, **_ignored):
        from .loading import save_pretrained
        save_pretrained(self, save_directory, dict(self.config.__dict__), type(self).__name__, variant)

    def set_adapters(self, adapter_names, weights=None):
        """``unet.set_adapters`` [EXT PeftAdapterMixin], /root/reference/utils/util.py:596"""
        from . import lora
        lora.set_adapters(self, adapter_names, weights)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    @property
    def attn_processors(self):
        return {n + ".processor": "lkgd_hip" for n, m in self.named_modules() if isinstance(m, Attention)}

    def set_attn_processor(self, processor):   # the HIP kernels are the only processor
        return None

    def set_default_attn_processor(self):
        return None

    def enable_forward_chunking(self, chunk_size=None, dim=0):
        if dim not in (0, 1):
            raise ValueError(f"Make sure to set `dim` to either 0 or 1, not {dim}")
        return None   # memory knob of the reference; a no-op at 288 GB

    def _set_gradient_checkpointing(self, module, value=False):
        return None

    # ---- weight packing ---------------------------------------------------------------------------------------
    def _register_temb(self, lin: nn.Linear) -> int:
        off = sum(l.out_features for l in self._temb_reg)
        self._temb_reg.append(lin)
        return off

    def _register_cross(self, attn2: Attention) -> int:
        off = sum(a.out_dim for a in self._cross_reg)
        self._cross_reg.append(attn2)
        return off

    def invalidate(self):
        """call after changing parameters in place (load_state_dict / .to() do it automatically)"""
        self._pk = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._pk = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._pk = None
        return r

    @torch.no_grad()
    def prepare(self):
        """(re)build the kernel-layout weights; idempotent, runs once per weight version"""
        if self._pk is not None:
            return
        if self.device.type != "cuda":
            raise LkgdHipError("lkgd_amd UNet runs on MI355X only: move the module to cuda before forward()")
        self._temb_reg, self._cross_reg = [], []
        for name, mod in self.named_modules():          # dotted paths for the roctx ranges (lkgd_amd/trace.py)
            mod._trace_name = name or type(self).__name__
        self.time_embedding.pack()
        self.add_embedding.pack()
        for blk in list(self.down_blocks) + [self.mid_block] + ([] if self._encoder_only else list(self.up_blocks)):
            blk.pack(self)
        pk = SimpleNamespace()
        pk.w_in = pack_conv3x3_c8(self.conv_in.weight.detach())
        pk.b_in = _f32(self.conv_in.bias)
        if not self._encoder_only:
            pk.gn_out = (_f32(self.conv_norm_out.weight), _f32(self.conv_norm_out.bias))
            pk.w_out = pack_conv3x3(self.conv_out.weight.detach())
            pk.b_out = _f32(self.conv_out.bias)
        self._pack_extra(pk)
        # all time_emb_proj layers of the model as ONE [sum C, 1280] GEMM per step
        pk.w_temb = torch.cat([pack_linear(l.weight) for l in self._temb_reg], dim=0).contiguous()
        pk.b_temb = torch.cat([_f32(l.bias) for l in self._temb_reg]).contiguous()
        # all one-token cross-attentions folded to ONE [sum C, 1024] GEMM per forward
        folded = [a.fold_cross() for a in self._cross_reg]
        pk.w_x = torch.cat([w for w, _ in folded], dim=0).to(torch.float16).contiguous()
        pk.b_x = torch.cat([b for _, b in folded]).contiguous()
        pk.x_var = {}                      # masked-LoRA variants of the folded cross-attention matrix (_cross_tables)
        from . import lora as _lora
        pk.has_lora = bool(_lora.lora_layers(self))
        self._pk = pk

    # ---- forward pieces ---------------------------------------------------------------------------------------
    def _time_embed(self, ctx: Ctx, timestep, added_time_ids) -> None:
        """unet_..._controlnet.py:388-426 -> table temb_all[b, :] = time_emb_proj_*(silu(emb[b]))"""
        dev, B = ctx.device, ctx.B
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32, device=dev)
        t = t.to(device=dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        boc0 = self.config.block_out_channels[0]
        emb = self.time_embedding.run(ops.timestep_embedding(t, boc0))
        ids = added_time_ids.to(device=dev, dtype=torch.float32).flatten().contiguous()
        if ids.numel() % B:
            raise ValueError("added_time_ids must be [batch, n_ids]")
        te = ops.timestep_embedding(ids, self.config.addition_time_embed_dim).reshape(B, -1)
        emb = self.add_embedding.run(te, res=emb)                  # emb + aug_emb
        semb = ops.silu(emb)
        pk = self._pk
        ctx.temb_all = torch.empty(B, pk.w_temb.shape[0], dtype=torch.float16, device=dev)
        ops.gemm(semb, pk.w_temb, ctx.temb_all, M=B, N=pk.w_temb.shape[0], K=pk.w_temb.shape[1], bias=pk.b_temb)

    def _cross_tables(self, ctx: Ctx, encoder_hidden_states: torch.Tensor) -> None:
        if encoder_hidden_states.dim() != 3:
            raise LkgdHipError("encoder_hidden_states must be [batch, tokens, cross_attention_dim]")
        pk = self._pk
        if encoder_hidden_states.shape[0] != ctx.B_total:
            raise ValueError(f"encoder_hidden_states carries {encoder_hidden_states.shape[0]} entries, expected "
                             f"{ctx.B_total}")
        Bt = ctx.B_total       # under CFG sharding every rank still needs ALL contexts (App. C11 interleaving)
        Lk = encoder_hidden_states.shape[1]
        if Lk != 1:
            # a multi-token context (never SVD's CLIP image token): attn2 runs as written, block by block (_cross_literal);
            # the folded-bias epilogues of the blocks then add a table of zeros
            if ctx.lora is not None:
                raise LkgdHipError("masked LoRA with a multi-token context is not supported")
            if Lk < 1 or Bt * Lk > 256:
                raise LkgdHipError(f"context of {Bt} x {Lk} tokens: lkgd_attn_cross holds at most 256 key rows")
            ctx.cross_Lk = Lk
            ctx.cross_e = encoder_hidden_states.to(device=ctx.device, dtype=torch.float16).reshape(Bt * Lk, -1).contiguous()
            ctx.xb_all = torch.zeros(Bt, pk.w_x.shape[0], dtype=torch.float16, device=ctx.device)
            return
        e = encoder_hidden_states.to(device=ctx.device, dtype=torch.float16).reshape(Bt, -1).contiguous()
        ctx.xb_all = torch.empty(Bt, pk.w_x.shape[0], dtype=torch.float16, device=ctx.device)
        ops.gemm(e, pk.w_x, ctx.xb_all, M=Bt, N=pk.w_x.shape[0], K=pk.w_x.shape[1], bias=pk.b_x)
        if ctx.lora is None:
            return
        # LoRA on attn2.to_v / to_out.0: one folded matrix per distinct adapter subset, one table per entry run (a run's
        # rows look up ANY context row - the temporal interleave of App. C11 - but always with the run's own weights)
        ctx.xb_runs = []
        for i in range(len(ctx.lora.runs)):
            key = tuple((ctx.lora.adapters(a.to_v, i), ctx.lora.adapters(a.to_out[0], i)) for a in self._cross_reg)
            if not any(k[0] or k[1] for k in key):
                ctx.xb_runs.append(ctx.xb_all)
                continue
            wx = pk.x_var.get(key)
            if wx is None:
                with _replay.invariant():
                    wx = torch.cat([a.fold_cross(k)[0] for a, k in zip(self._cross_reg, key)], dim=0)
                    wx = pk.x_var[key] = wx.to(torch.float16).contiguous()
            tab = torch.empty_like(ctx.xb_all)
            ops.gemm(e, wx, tab, M=Bt, N=wx.shape[0], K=wx.shape[1], bias=pk.b_x)
            ctx.xb_runs.append(tab)

    def _joint_maps(self, ctx: Ctx):
        """partner permutations of the patch API (patch/patch.py:454-475), computed on the host from the 4-entry mask"""
        mask = getattr(self, "_joint_attn_mask", None)
        if mask is None:
            return
        info = getattr(self, "_tome_info", None)
        flip = bool(info and info["args"].get("flip", False))
        if ctx.shard is not None:
            # a sharded rank holds the batch entries [b0, b0 + B) of the call: the mask of the whole UNet batch is cut to them.  The
            # reference pairs the i-th unmasked with the i-th masked entry of the WHOLE batch (patch.py:466-468); that pairing
            # stays inside the rank when everything in front of its slice and the slice itself hold as many masked as unmasked
            # entries - true for the loader's [0,1,0,1] over [u_x, u_y, c_x, c_y] split into CFG halves
            full = [bool(v) for v in torch.as_tensor(mask).tolist()]
            if ctx.B_total % len(full):
                raise LkgdHipError("joint_attn_mask length must divide the UNet batch")
            full = [v for v in full for _ in range(ctx.B_total // len(full))]
            head, own = full[:ctx.b0], full[ctx.b0:ctx.b0 + ctx.B]
            if sum(head) * 2 != len(head) or sum(own) * 2 != len(own):
                raise LkgdHipError("joint attention under sharding: a rank's batch entries must hold whole (masked, unmasked) "
                                   f"pairs; mask {full}, entries {ctx.b0}..{ctx.b0 + ctx.B - 1}")
            if flip and ctx.frames_sharded:
                # frame f of an entry attends to frame F-1-f of its partner: on the MIRROR shard, at local position f_local-1-t when
                # the slices are symmetric.  The spatial joint block then reads its K | V rows from the mirror shard's buffer
                # (BasicTransformerBlock._joint), and the partner map below - built on LOCAL frame counts - indexes that buffer
                pl = ctx.shard.plan
                k_, si_ = pl.frame_shards, pl.shard_index
                if pl.splits[si_] != pl.splits[k_ - 1 - si_] or pl.f0 + pl.f_local != pl.num_frames - sum(pl.splits[:k_ - 1 - si_]):
                    raise LkgdHipError("joint attention with flip=True under frame sharding needs symmetric frame slices "
                                       f"(dist.make_plan(symmetric=True)); this plan's are {pl.splits}")
                ctx.flip_mirror = True
            mask = own

        def partner(n_rows, group):
            m = torch.as_tensor(mask, dtype=torch.bool).repeat_interleave(n_rows // len(mask))
            idx = torch.arange(n_rows)
            p = torch.empty(n_rows, dtype=torch.long)
            p[~m] = idx[m]
            p[m] = idx[~m]
            if group is not None:
                p = p.reshape(-1, group).flip(1).reshape(-1)
            return p.to(torch.int32)
        if ctx.N % len(mask) or ctx.B % len(mask):
            raise LkgdHipError("joint_attn_mask length must divide the UNet batch")
        ctx.spatial_partner = partner(ctx.N, ctx.F if flip else None).to(ctx.device)
        tp = partner(ctx.B, None)
        ctx.entry_partner = tp.tolist()                       # batch-entry level partner (host copy, for the LoRA plan)
        ctx.temporal_partner = tp.to(ctx.device)
        # entry blocks for post == "conv_fuse": i-th masked block <-> i-th unmasked block (no flip, patch.py:488-493)
        ml = [bool(v) for v in torch.as_tensor(mask).tolist()]
        per = ctx.N // len(ml)
        on, off = [j for j, v in enumerate(ml) if v], [j for j, v in enumerate(ml) if not v]
        ctx.joint_blocks = None
        if len(on) == len(off):
            pj = {a: b for a, b in zip(on, off)}
            pj.update({b: a for a, b in zip(on, off)})
            ctx.joint_blocks = [(j * per, per, pj[j] * per, ml[j]) for j in range(len(ml))]

    def _run(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals,
             mid_block_additional_residual, added_time_ids):
        """reference forward body (unet_..._controlnet.py:388-503) from the NCHW API tensors"""
        if sample.dim() != 5:
            raise ValueError("sample must be [batch, frames, channels, height, width]")
        B, F, Cin, H, W = sample.shape
        x = sample.to(device=self.device, dtype=torch.float16).reshape(B * F, Cin, H, W).contiguous()
        tok = ops.nchw_to_tokens(x)
        return self.forward_tokens(tok, B, F, H, W, timestep, encoder_hidden_states, added_time_ids,
                                   down_block_additional_residuals, mid_block_additional_residual)

    def _res_tokens(self, add: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        if add.dim() == 2:          # already a channels-last token matrix (lkgd_amd.controlnet's token-level output)
            if add.shape != like.shape or add.dtype != torch.float16 or add.device != like.device:
                raise ValueError(f"additional residual tokens {tuple(add.shape)} do not match skip {tuple(like.shape)}")
            return add
        a = add.to(device=like.device, dtype=torch.float16).contiguous()
        t = ops.nchw_to_tokens(a)
        if t.shape != like.shape:
            raise ValueError(f"additional residual of shape {tuple(add.shape)} does not match skip {tuple(like.shape)}")
        return t

    @torch.no_grad()
    def forward_tokens(self, tokens: torch.Tensor, B: int, F: int, H: int, W: int, timestep, encoder_hidden_states,
                       added_time_ids, down_block_additional_residuals=None, mid_block_additional_residual=None,
                       shard=None):
        """channels-last entry: input tokens [B*F*H*W, 8] -> (noise tokens [B*F*H*W, 4], ctx).  This is what
        lkgd_amd.pipeline calls between the glue kernels (no NCHW conversions inside the loop)."""
        with _replay.invariant():          # (a first call after invalidate() packs the weights: constants of a plan being recorded)
            self.prepare()
        with _trace.range_("unet_forward"):
            return self._forward_tokens(tokens, B, F, H, W, timestep, encoder_hidden_states, added_time_ids,
                                        down_block_additional_residuals, mid_block_additional_residual, shard)

    def _forward_tokens(self, tokens, B, F, H, W, timestep, encoder_hidden_states, added_time_ids,
                        down_block_additional_residuals, mid_block_additional_residual, shard):
        ctx = Ctx(B, F, H, W, self.device, shard)
        pk = self._pk
        self._time_embed(ctx, timestep, added_time_ids)
        self._joint_maps(ctx)
        if pk.has_lora:
            from . import lora as _lora
            ctx.lora = _lora.entry_plan(self, ctx.B, ctx.entry_partner, ctx.B_total, ctx.b0)
        self._cross_tables(ctx, encoder_hidden_states)
        h = ctx.new(ctx.T, pk.w_in.shape[0])
        ops.gemm(tokens, pk.w_in, h, M=ctx.T, N=pk.w_in.shape[0], K=128, bias=pk.b_in, mode=ops.A_CONV3X3_C8, Cin=8,
                 conv=(H, W, H, W, 1, 0))
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk.run(ctx, h)
            skips += [o for o, _, _ in outs]
            if down_block_additional_residuals is not None:
                # the reference adds the ControlNet residuals INSIDE the block loop and `zip` truncates, so early
                # skips receive their residual once per remaining down block (App. C1) - reproduced as is
                skips = [ops.add(s, self._res_tokens(add, s))
                         for s, add in zip(skips, down_block_additional_residuals)]
        h = self.mid_block.run(ctx, h)
        if mid_block_additional_residual is not None:
            h = ops.add(h, self._res_tokens(mid_block_additional_residual, h))
        for blk in self.up_blocks:
            k = len(blk.resnets)
            part, skips = skips[-k:], skips[:-k]
            h = blk.run(ctx, h, list(part))      # .pop() takes from the end, as the reference does
        n = ops.groupnorm_silu(h, None, ctx.N, ctx.HW, *pk.gn_out, 1e-5)
        co, c0 = self.config.out_channels, self.config.block_out_channels[0]
        out_tok = ctx.new(ctx.T, co)
        ops.gemm(n, pk.w_out, out_tok, M=ctx.T, N=co, K=9 * c0, bias=pk.b_out, mode=ops.A_CONV3X3, Cin=c0,
                 conv=(ctx.H, ctx.W, ctx.H, ctx.W, 1, 0))
        return out_tok, ctx

    def _finish(self, out_tok, ctx, B, F, return_dict):
        co = self.config.out_channels
        out = ops.tokens_to_nchw(out_tok, B * F, co, ctx.H, ctx.W).reshape(B, F, co, ctx.H, ctx.W)
        if not return_dict:
            return (out,)
        return UNetSpatioTemporalConditionOutput(sample=out)


class UNetSpatioTemporalConditionControlNetModel(_UNetBase):
    """stock signature - reference models/unet_spatio_temporal_condition_controlnet.py:358-368"""

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor,
                down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
                mid_block_additional_residual: Optional[torch.Tensor] = None, return_dict: bool = True,
                added_time_ids: torch.Tensor = None):
        out_tok, ctx = self._run(sample, timestep, encoder_hidden_states, down_block_additional_residuals,
                                 mid_block_additional_residual, added_time_ids)
        return self._finish(out_tok, ctx, sample.shape[0], sample.shape[1], return_dict)


class QuaternionLinearAutograd(nn.Module):
    """parameter holder for core_qnn's layer (r/i/j/k weights [in/4, out/4] + bias), SURVEY.md App. A.8"""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        i4, o4 = in_features // 4, out_features // 4
        s = 1.0 / math.sqrt(in_features)
        self.r_weight = nn.Parameter(torch.randn(i4, o4) * s)
        self.i_weight = nn.Parameter(torch.randn(i4, o4) * s)
        self.j_weight = nn.Parameter(torch.randn(i4, o4) * s)
        self.k_weight = nn.Parameter(torch.randn(i4, o4) * s)
        self.bias = nn.Parameter(torch.zeros(out_features))


class UNetSpatioTemporalConditionModel(_UNetBase):
    """LKGD signature - reference models/unet_spatio_temporal_condition.py:448-459; ``domain_features`` and
    ``flow_features`` are positional.  The latent-knowledge fuse (:536-595) is step-invariant, so it is evaluated once
    per distinct (embedding, domain, flow) triple and cached (lkgd_amd/lk_fuse.py)."""

    def _init_extra(self, cfg):
        def dw():
            return nn.Conv1d(1024, 256, kernel_size=1, groups=256, bias=False)
        self.quaternion_lora_dconv = dw()
        self.quaternion_lora_lconv = dw()
        self.quaternion_lora_fconv = dw()
        self.quaternion_lora_fuse = QuaternionLinearAutograd(1024, 512)
        self.quaternion_lora_fuse_fft_mag = QuaternionLinearAutograd(512, 256)
        self.quaternion_lora_fuse_fft_pha = QuaternionLinearAutograd(512, 256)
        self.quaternion_lora_fuse_fft_mag0 = nn.Linear(4, 1)
        self.quaternion_lora_fuse_fft_pha0 = nn.Linear(4, 1)
        self.quaternion_lora_fuse_sf = nn.Sequential(nn.Linear(1024, 256), nn.LeakyReLU(0.1, inplace=True),
                                                     nn.Linear(256, 1024))
        self.quaternion_lora_texts = nn.Parameter(torch.zeros(256))
        self.quaternion_lora_texts_fft_mag = nn.Parameter(torch.zeros(129))
        self.quaternion_lora_texts_fft_pha = nn.Parameter(torch.zeros(129))
        self._lk_cache = None

    def invalidate(self):
        super().invalidate()
        self._lk_cache = None

    def fused_embedding(self, encoder_hidden_states, domain_features, flow_features) -> torch.Tensor:
        from .lk_fuse import lk_fuse_cached
        return lk_fuse_cached(self, encoder_hidden_states, domain_features, flow_features)

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, domain_features, flow_features,
                down_block_additional_residuals=None, mid_block_additional_residual=None, return_dict: bool = True,
                added_time_ids=None):
        enc = self.fused_embedding(encoder_hidden_states, domain_features, flow_features)
        out_tok, ctx = self._run(sample, timestep, enc, down_block_additional_residuals,
                                 mid_block_additional_residual, added_time_ids)
        return self._finish(out_tok, ctx, sample.shape[0], sample.shape[1], return_dict)


def init_synthetic_weights_(model: nn.Module, seed: int = 0) -> nn.Module:
    """random-init for benchmarking on the device the model lives on (there are no checkpoints on the GPU box):
    zero-mean fan-in normal weights, norm affine ~ (1, 0) with small noise, small biases."""
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)

    def rn(p, scale=1.0, shift=0.0):
        p.copy_((torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) * scale + shift).to(p.dtype))

    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                rn(m.weight, 0.1, 1.0)
                rn(m.bias, 0.1)
            elif isinstance(m, (nn.Linear, nn.Conv1d, nn.Conv2d, nn.Conv3d)):
                rn(m.weight, 1.0 / m.weight[0].numel() ** 0.5)
                if m.bias is not None:
                    rn(m.bias, 0.02)
            elif isinstance(m, QuaternionLinearAutograd):
                for w in (m.r_weight, m.i_weight, m.j_weight, m.k_weight):
                    rn(w, 1.0 / (w.shape[0] * 4) ** 0.5)
                rn(m.bias, 0.02)
        for name, p in model.named_parameters():
            if name.endswith("mix_factor") or name.startswith("quaternion_lora_texts"):
                rn(p)
    if hasattr(model, "invalidate"):
        model.invalidate()
    return model
