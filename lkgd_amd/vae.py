"""The SVD VAE (``AutoencoderKLTemporalDecoder``) on the MI355X path: encode of the conditioning image, temporal decode of
the denoised latents (SURVEY.md 8f rank 2 - the clip-level stages either side of the Euler loop).

Call sites in the reference: /root/reference/pipeline/pipeline_stable_video_diffusion_trans.py:205-226
(``_encode_vae_image``: ``vae.encode(image).latent_dist.mode()``), :256-283 (``decode_latents``:
``vae.decode(latents[i:i+chunk], num_frames=chunk).sample`` in chunks of ``decode_chunk_size``), :470-484 / :643-645 (the
``force_upcast`` dance).  The class itself is diffusers 0.27.2 [EXT] - module tree and parameter names are restated so a real
``vae/`` checkpoint loads with ``load_state_dict`` / ``from_pretrained`` (oracle/vae.py carries the same restatement in fp32
torch for the tests; PARITY UNPINNED: nothing under /root/reference holds VAE code or fixtures).

MI355X design: the same channels-last fp16 token matrices and the same kernels as the UNet -
* every 3x3 convolution (128..512 channels, also the 2x nearest upsample and the encoder's asymmetric stride-2 downsample) is
  the implicit-GEMM kernel (``lkgd_gemm_f16``; ``pad_off`` = the encoder's F.pad(0,1,0,1) + padding-0 conv); the 3 / 4
  channel inputs go through the 8-channel-padded conv_in paths; ``conv_out`` (-> 3 or 8 channels) is a GEMM with the output
  channels padded to 8, and the encoder's 1x1 ``quant_conv`` is folded into its ``conv_out`` (exact algebra);
* GroupNorm(32) + SiLU, spatial and temporal (statistics across the frames of the chunk), are the UNet's norm kernels;
* temporal Conv3d (3,1,1) = the frame-shifted implicit GEMM; AlphaBlender("learned", switch) is its epilogue
  (``s + sigmoid(mix) * conv2(..)``);
* the mid-block attention is ONE head of dim 512 over 72x128 = 9216 tokens per frame: QK^T and PV are plain GEMMs on the
  matrix cores ([9216 x 9216] fp16 scores per frame, 170 MB, reused), ``lkgd_softmax_rows`` between them; V is produced
  transposed by swapping the GEMM operands (V^T = W_v x^T) and its bias is added after PV (softmax rows sum to 1);
* ``time_conv_out`` + the NCHW conversion of the decoded frames is one small kernel.
Activations are fp16 with fp32 accumulation whatever dtype the module's parameters are kept in (``.to(torch.float32)`` of
the ``force_upcast`` path changes the holders, not the kernels).  288 GB: a whole 14-frame 576x1024 chunk decodes at once
(largest activation 2.1 GB); ``decode_chunk_size`` is honoured because it changes the temporal statistics, not for memory.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import LkgdHipError
from .packing import pack_conv3x3, pack_conv3x3_c8, pack_linear, pack_tconv3


@dataclass
class VAEConfig:
    """constructor keywords of AutoencoderKLTemporalDecoder [EXT] (SVD ``vae/config.json``)"""
    in_channels: int = 3
    out_channels: int = 3
    down_block_types: Tuple[str, ...] = ("DownEncoderBlock2D",) * 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 4
    sample_size: int = 768
    scaling_factor: float = 0.18215
    force_upcast: bool = True


def _f32(p):
    return p.detach().to(torch.float32).contiguous()


def _gn(x, nsamples, rows, norm: nn.GroupNorm, silu=True):
    return ops.groupnorm_silu(x, None, nsamples, rows, norm._g, norm._b, norm.eps, silu=silu)


def _pack_gn(norm: nn.GroupNorm):
    norm._g, norm._b = _f32(norm.weight), _f32(norm.bias)


def _new(rows, cols, dev):
    return torch.empty(rows, cols, dtype=torch.float16, device=dev)


# ------------------------------------------------------------------------------------------------ parameter holders
class ResnetBlock2D(nn.Module):
    """ResnetBlock2D(temb_channels=None, groups=32) [EXT resnet.py]"""

    def __init__(self, cin: int, cout: int, eps: float = 1e-6):
        super().__init__()
        if cin % 64 or cout % 64:
            raise LkgdHipError("VAE channel counts must be multiples of 64 (implicit-GEMM K granularity)")
        self.cin, self.cout = cin, cout
        self.norm1 = nn.GroupNorm(32, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def pack(self):
        _pack_gn(self.norm1); _pack_gn(self.norm2)
        self._w1, self._b1 = pack_conv3x3(self.conv1.weight.detach()), _f32(self.conv1.bias)
        self._w2, self._b2 = pack_conv3x3(self.conv2.weight.detach()), _f32(self.conv2.bias)
        if self.conv_shortcut is not None:
            self._ws, self._bs = pack_linear(self.conv_shortcut.weight.detach()), _f32(self.conv_shortcut.bias)

    def run(self, x, N, H, W, next_rows=None):
        """``next_rows``: rows per sample of the GroupNorm that reads this block's output (default H*W: a spatial norm; the
        temporal res block's norm spans F*H*W) - the convolutions leave that norm's column sums where their tile program
        has them (ops.gemm ``colstats``: the 256 / 512-channel levels), and its separate read pass disappears"""
        T, geo = N * H * W, (H, W, H, W, 1, 0)
        n1 = _gn(x, N, H * W, self.norm1)
        h = _new(T, self.cout, x.device)
        ops.gemm(n1, self._w1, h, M=T, N=self.cout, K=9 * self.cin, bias=self._b1, mode=ops.A_CONV3X3, Cin=self.cin, conv=geo,
                 colstats=H * W)
        n2 = _gn(h, N, H * W, self.norm2)
        sc = x
        if self.conv_shortcut is not None:
            sc = _new(T, self.cout, x.device)
            ops.gemm(x, self._ws, sc, M=T, N=self.cout, K=self.cin, bias=self._bs)
        out = _new(T, self.cout, x.device)
        ops.gemm(n2, self._w2, out, M=T, N=self.cout, K=9 * self.cout, bias=self._b2, mode=ops.A_CONV3X3, Cin=self.cout,
                 conv=geo, res1=sc, colstats=next_rows or H * W)
        return out


class TemporalResnetBlock(nn.Module):
    def __init__(self, c: int, eps: float = 1e-5):
        super().__init__()
        self.c = c
        self.norm1 = nn.GroupNorm(32, c, eps=eps)
        self.conv1 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))
        self.norm2 = nn.GroupNorm(32, c, eps=eps)
        self.conv2 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))

    def pack(self):
        _pack_gn(self.norm1); _pack_gn(self.norm2)
        self._w1, self._b1 = pack_tconv3(self.conv1.weight.detach()), _f32(self.conv1.bias)
        self._w2, self._b2 = pack_tconv3(self.conv2.weight.detach()), _f32(self.conv2.bias)


class AlphaBlender(nn.Module):
    def __init__(self, alpha: float = 0.0):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([alpha], dtype=torch.float32))


class SpatioTemporalResBlock(nn.Module):
    """SpatioTemporalResBlock(temb_channels=None, merge_strategy="learned", switch_spatial_to_temporal_mix=True) [EXT]"""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(cin, cout, 1e-6)
        self.temporal_res_block = TemporalResnetBlock(cout, 1e-5)
        self.time_mixer = AlphaBlender(0.0)

    def pack(self):
        self.spatial_res_block.pack()
        self.temporal_res_block.pack()
        # alpha = 1 - sigmoid(mix): out = alpha*s + (1-alpha)*(s + conv2(..)) = s + sigmoid(mix) * conv2(..)
        self._wt = float(torch.sigmoid(self.time_mixer.mix_factor.detach().float()).item())

    def run(self, x, B, F, H, W):
        s = self.spatial_res_block.run(x, B * F, H, W, next_rows=F * H * W)
        t, c, HW, T = self.temporal_res_block, self.temporal_res_block.c, H * W, B * F * H * W
        n3 = _gn(s, B, F * HW, t.norm1)                      # statistics span the frames of the chunk
        h = _new(T, c, x.device)
        ops.gemm(n3, t._w1, h, M=T, N=c, K=3 * c, bias=t._b1, mode=ops.A_TCONV3, Cin=c, tconv=(F, HW), colstats=F * HW)
        n4 = _gn(h, B, F * HW, t.norm2)
        out = _new(T, c, x.device)
        ops.gemm(n4, t._w2, out, M=T, N=c, K=3 * c, bias=t._b2, mode=ops.A_TCONV3, Cin=c, tconv=(F, HW), s_acc=self._wt,
                 res1=s, colstats=HW)              # -> the next block's spatial norm1 / conv_norm_out
        return out


class Attention(nn.Module):
    """Attention(query_dim=C, heads=C // head_dim, dim_head=head_dim, norm_num_groups=32, eps=1e-6, bias=True,
    residual_connection=True) [EXT attention_processor.py]; the VAE uses ONE head of dim C"""

    def __init__(self, c: int, head_dim: int):
        super().__init__()
        if c != head_dim:
            raise LkgdHipError("VAE attention: one head of dim C (attention_head_dim == channels) is the only SVD configuration")
        self.c = c
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def pack(self):
        _pack_gn(self.group_norm)
        self._wq, self._bq = pack_linear(self.to_q.weight.detach()), _f32(self.to_q.bias)
        self._wk, self._bk = pack_linear(self.to_k.weight.detach()), _f32(self.to_k.bias)
        self._wv, self._bv = pack_linear(self.to_v.weight.detach()), _f32(self.to_v.bias)
        self._wo, self._bo = pack_linear(self.to_out[0].weight.detach()), _f32(self.to_out[0].bias)

    def run(self, x, N, H, W):
        c, S, dev = self.c, H * W, x.device
        if S % 8:
            raise LkgdHipError("VAE attention needs H*W % 8 == 0")
        Sp = (S + 63) // 64 * 64                                 # K granularity of the PV product
        T = N * S
        xn = _gn(x, N, S, self.group_norm, silu=False)
        q, k = _new(T, c, dev), _new(T, c, dev)
        ops.gemm(xn, self._wq, q, M=T, N=c, K=c, bias=self._bq)
        ops.gemm(xn, self._wk, k, M=T, N=c, K=c, bias=self._bk)
        o = _new(T, c, dev)
        pad = Sp != S
        sc = (torch.zeros if pad else torch.empty)(S, Sp, dtype=torch.float16, device=dev)      # one frame's scores, reused
        vT = (torch.zeros if pad else torch.empty)(c, Sp, dtype=torch.float16, device=dev)
        scale = float(c) ** -0.5
        for n in range(N):
            r = slice(n * S, (n + 1) * S)
            ops.gemm(self._wv, xn[r], vT, M=c, N=S, K=c)                       # V^T = W_v x^T (operands swapped; no bias)
            ops.gemm(q[r], k[r], sc, M=S, N=S, K=c, s_acc=scale)               # scores = scale * Q K^T
            ops.softmax_rows(sc[:, :S])
            ops.gemm(sc, vT, o[r], M=S, N=c, K=Sp, bias=self._bv)              # P V + b_v (rows of P sum to 1)
        out = _new(T, c, dev)
        ops.gemm(o, self._wo, out, M=T, N=c, K=c, bias=self._bo, res1=x)       # to_out + residual connection
        return out


class Downsample2D(nn.Module):
    """Downsample2D(padding=0): F.pad(x, (0,1,0,1)) + 3x3 stride-2 conv [EXT downsampling.py]"""

    def __init__(self, c: int):
        super().__init__()
        self.c = c
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def pack(self):
        self._w, self._b = pack_conv3x3(self.conv.weight.detach()), _f32(self.conv.bias)

    def run(self, x, N, H, W):
        Ho, Wo = (H - 2) // 2 + 1, (W - 2) // 2 + 1
        out = _new(N * Ho * Wo, self.c, x.device)
        ops.gemm(x, self._w, out, M=N * Ho * Wo, N=self.c, K=9 * self.c, bias=self._b, mode=ops.A_CONV3X3, Cin=self.c,
                 conv=(Ho, Wo, H, W, 2, 0, 1))
        return out, Ho, Wo


class Upsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.c = c
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def pack(self):
        self._w, self._b = pack_conv3x3(self.conv.weight.detach()), _f32(self.conv.bias)

    def run(self, x, N, H, W):
        Ho, Wo = 2 * H, 2 * W
        out = _new(N * Ho * Wo, self.c, x.device)
        ops.gemm(x, self._w, out, M=N * Ho * Wo, N=self.c, K=9 * self.c, bias=self._b, mode=ops.A_CONV3X3, Cin=self.c,
                 conv=(Ho, Wo, H, W, 1, 1), colstats=Ho * Wo)     # nearest-2x folded into the gather
        return out, Ho, Wo


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None


class UNetMidBlock2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(c, c)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c), ResnetBlock2D(c, c)])


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc = cfg.block_out_channels
        if cfg.in_channels > 8 or boc[0] % 16:
            raise LkgdHipError("VAE encoder: at most 8 input channels, block_out_channels[0] % 16 == 0")
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, co in enumerate(boc):
            self.down_blocks.append(DownEncoderBlock2D(c, co, cfg.layers_per_block, i != len(boc) - 1))
            c = co
        self.mid_block = UNetMidBlock2D(boc[-1])
        self.conv_norm_out = nn.GroupNorm(32, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)


class MidBlockTemporalDecoder(nn.Module):
    def __init__(self, c, layers):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(c, c) for _ in range(layers)])
        self.attentions = nn.ModuleList([Attention(c, c)])


class UpBlockTemporalDecoder(nn.Module):
    def __init__(self, cin, cout, layers, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(cin if i == 0 else cout, cout) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None


class TemporalDecoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc = cfg.block_out_channels
        if cfg.latent_channels > 8 or cfg.out_channels != 3:
            raise LkgdHipError("VAE decoder: at most 8 latent channels, 3 output channels")
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = MidBlockTemporalDecoder(boc[-1], cfg.layers_per_block)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(UpBlockTemporalDecoder(c, co, cfg.layers_per_block + 1, i != len(boc) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(cfg.out_channels, cfg.out_channels, (3, 1, 1), padding=(1, 0, 0))


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class _Dist:
    """DiagonalGaussianDistribution as far as the pipelines use it: ``mode()`` / ``mean``"""

    def __init__(self, mean):
        self.mean = mean

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        raise LkgdHipError("latent_dist.sample() is not used by the SVD pipelines (they take .mode())")


@dataclass
class AutoencoderKLOutput:
    latent_dist: _Dist


def _pad_cout(w: torch.Tensor, b: torch.Tensor, n: int):
    """pad the output channels of a conv to n (zero weights / biases): GEMM output granularity"""
    if w.shape[0] >= n:
        return w, b
    wp = torch.zeros((n,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    wp[:w.shape[0]] = w
    bp = torch.zeros(n, dtype=b.dtype, device=b.device)
    bp[:b.shape[0]] = b
    return wp, bp


class AutoencoderKLTemporalDecoder(nn.Module):
    def __init__(self, config: Optional[VAEConfig] = None, **kw):
        super().__init__()
        cfg = config if config is not None else VAEConfig(**kw)
        self.config = SimpleNamespace(**cfg.__dict__)
        self.encoder = Encoder(cfg)
        self.decoder = TemporalDecoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self._packed = False

    # ---- reference API surface ---------------------------------------------------------------------------------
    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    @property
    def device(self):
        return self.quant_conv.weight.device

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: Optional[str] = None, torch_dtype=None,
                        variant: Optional[str] = None, **_ignored):
        from .loading import build_from_pretrained
        return build_from_pretrained(cls, VAEConfig, pretrained_model_name_or_path, subfolder, torch_dtype, variant)

    def save_pretrained(self, save_directory: str, variant: Optional[str] = None, **_ignored):
        from .loading import save_pretrained
        save_pretrained(self, save_directory, dict(self.config.__dict__), type(self).__name__, variant)

    def invalidate(self):
        self._packed = False

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._packed = False
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._packed = False
        return r

    # ---- packing -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def prepare(self):
        if self._packed:
            return
        if self.device.type != "cuda":
            raise LkgdHipError("lkgd_amd VAE runs on MI355X only: move the module to cuda first")
        for m in self.modules():
            if isinstance(m, (SpatioTemporalResBlock, Attention, Downsample2D, Upsample2D)):
                m.pack()
        for blk in list(self.encoder.down_blocks) + [self.encoder.mid_block]:
            for r in blk.resnets:
                r.pack()
        e, d, lc = self.encoder, self.decoder, self.config.latent_channels
        # encoder conv_in: direct small-channel kernel (3 -> 8 padded input channels)
        w = e.conv_in.weight.detach().permute(0, 2, 3, 1)
        w = torch.nn.functional.pad(w, (0, 8 - w.shape[3]))
        e._w_in, e._b_in = w.to(torch.float16).contiguous(), _f32(e.conv_in.bias)
        _pack_gn(e.conv_norm_out)
        # quant_conv (1x1) folded into conv_out: W' = Wq . Wc (per tap), b' = Wq bc + bq
        wq = self.quant_conv.weight.detach().float().reshape(2 * lc, 2 * lc)
        wc, bc = e.conv_out.weight.detach().float(), e.conv_out.bias.detach().float()
        wf = torch.einsum("oq,qckl->ockl", wq, wc)
        bf = wq @ bc + self.quant_conv.bias.detach().float()
        wf, bf = _pad_cout(wf, bf, 8)
        e._w_out, e._b_out = pack_conv3x3(wf), bf.contiguous()
        # decoder conv_in: the 8-channel-padded implicit-GEMM path
        w = d.conv_in.weight.detach()
        wp = torch.zeros(w.shape[0], 8, 3, 3, dtype=w.dtype, device=w.device)
        wp[:, :w.shape[1]] = w
        d._w_in, d._b_in = pack_conv3x3_c8(wp), _f32(d.conv_in.bias)
        _pack_gn(d.conv_norm_out)
        wo, bo = _pad_cout(d.conv_out.weight.detach().float(), d.conv_out.bias.detach().float(), 8)
        d._w_out, d._b_out = pack_conv3x3(wo), bo.contiguous()
        d._w_t = _f32(d.time_conv_out.weight.detach().reshape(3, 3, 3))          # [co][ci][kt]
        d._b_t = _f32(d.time_conv_out.bias)
        self._packed = True

    # ---- encode ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """[N, 3, H, W] in [-1, 1] -> latent_dist with ``mode()`` = mean [N, 4, H/8, W/8] (in x's dtype)"""
        self.prepare()
        if x.dim() != 4 or x.shape[1] != self.config.in_channels:
            raise ValueError(f"expected [N, {self.config.in_channels}, H, W]")
        N, C_, H, W = x.shape
        down = 2 ** (len(self.config.block_out_channels) - 1)
        if H % down or W % down:
            raise ValueError(f"height and width must be divisible by {down}")
        dev, e = self.device, self.encoder
        xh = x.to(device=dev, dtype=torch.float16).contiguous()
        tok = torch.zeros(N * H * W, 8, dtype=torch.float16, device=dev)
        ops.nchw_to_tokens(xh, out=tok[:, :C_])
        h = ops.conv3x3_small(tok, e._w_in, e._b_in, N, H, W, 1, False)
        for blk in e.down_blocks:
            for r in blk.resnets:
                h = r.run(h, N, H, W)
            if blk.downsamplers is not None:
                h, H, W = blk.downsamplers[0].run(h, N, H, W)
        m = e.mid_block
        h = m.resnets[0].run(h, N, H, W)
        h = m.attentions[0].run(h, N, H, W)
        h = m.resnets[1].run(h, N, H, W)
        n = _gn(h, N, H * W, e.conv_norm_out)
        c = self.config.block_out_channels[-1]
        mom = _new(N * H * W, 8, dev)
        ops.gemm(n, e._w_out, mom, M=N * H * W, N=8, K=9 * c, bias=e._b_out, mode=ops.A_CONV3X3, Cin=c,
                 conv=(H, W, H, W, 1, 0))
        lc = self.config.latent_channels
        mean = ops.tokens_to_nchw(mom, N, lc, H, W).to(x.dtype)       # first `latent_channels` columns = the mean
        if not return_dict:
            return (_Dist(mean),)
        return AutoencoderKLOutput(latent_dist=_Dist(mean))

    # ---- decode ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z: torch.Tensor, num_frames: int = 1, return_dict: bool = True):
        """[B*num_frames, 4, h, w] -> sample [B*num_frames, 3, 8h, 8w] (in z's dtype)"""
        self.prepare()
        if z.dim() != 4 or z.shape[1] != self.config.latent_channels:
            raise ValueError(f"expected [batch*frames, {self.config.latent_channels}, h, w]")
        BF, lc, H, W = z.shape
        if num_frames <= 0 or BF % num_frames:
            raise ValueError("batch*frames must be divisible by num_frames")
        B, F = BF // num_frames, num_frames
        dev, d = self.device, self.decoder
        zh = z.to(device=dev, dtype=torch.float16).contiguous()
        tok = torch.zeros(BF * H * W, 8, dtype=torch.float16, device=dev)
        ops.nchw_to_tokens(zh, out=tok[:, :lc])
        c = self.config.block_out_channels[-1]
        h = _new(BF * H * W, c, dev)
        ops.gemm(tok, d._w_in, h, M=BF * H * W, N=c, K=128, bias=d._b_in, mode=ops.A_CONV3X3_C8, Cin=8, conv=(H, W, H, W, 1, 0))
        m = d.mid_block
        h = m.resnets[0].run(h, B, F, H, W)
        for r, a in zip(m.resnets[1:], m.attentions):
            h = r.run(a.run(h, BF, H, W), B, F, H, W)
        for blk in d.up_blocks:
            for r in blk.resnets:
                h = r.run(h, B, F, H, W)
            if blk.upsamplers is not None:
                h, H, W = blk.upsamplers[0].run(h, BF, H, W)
        n = _gn(h, BF, H * W, d.conv_norm_out)
        c0 = self.config.block_out_channels[0]
        rgb = _new(BF * H * W, 8, dev)
        ops.gemm(n, d._w_out, rgb, M=BF * H * W, N=8, K=9 * c0, bias=d._b_out, mode=ops.A_CONV3X3, Cin=c0,
                 conv=(H, W, H, W, 1, 0))
        out_dtype = z.dtype if z.dtype in (torch.float16, torch.float32) else torch.float32
        out = ops.time_conv_out(rgb, d._w_t, d._b_t, B, F, H, W, dtype=out_dtype)
        if not return_dict:
            return (out,)
        return DecoderOutput(sample=out)

    def forward(self, sample, num_frames: int = 1):
        z = self.encode(sample).latent_dist.mode()
        return self.decode(z, num_frames=num_frames)
