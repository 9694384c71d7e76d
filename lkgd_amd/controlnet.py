"""ControlNet-SVD encoder on the MI355X path.

Mirrors /root/reference/models/controlnet_sdv.py: ``ControlNetConditioningEmbeddingSVD`` :64-119,
``ControlNetSDVModel.__init__`` :156-317 (same parameter tree and state-dict key names: ``conv_in``, ``time_embedding``,
``add_embedding``, ``down_blocks``, ``controlnet_down_blocks.N``, ``controlnet_cond_embedding.{conv_in,blocks.N,conv_out}``,
``controlnet_mid_block``, ``mid_block``) and ``forward`` :441-578 (same signature and return value).

The encoder is the UNet's own (``lkgd_amd.unet._UNetBase`` with ``_encoder_only``): conv_in, embeddings, the four down
blocks and the mid block run the same HIP kernels.  On top of it:

* the conditioning embedding - eight 3x3 convolutions on the PIXEL grid at 3..256 channels - does not depend on the
  denoising step, so it is evaluated once per conditioning tensor and cached (the reference re-runs it every step).
  All layers but the last use the direct small-channel kernel
  (``lkgd_conv3x3_small``); ``conv_out`` (256 -> 320) is an implicit-GEMM convolution whose epilogue adds ``conv_in``'s
  output, i.e. ``sample + controlnet_cond`` costs nothing;
* the thirteen zero-initialised 1x1 convolutions are GEMMs with ``conditioning_scale`` folded into the epilogue scale.

``forward`` returns NCHW tensors like the reference; ``forward_tokens`` returns the channels-last token matrices, which
``lkgd_amd.unet`` accepts directly as ``down_block_additional_residuals`` (no layout conversions inside the loop).
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import LkgdHipError
from .packing import pack_conv3x3, pack_linear
from .unet import Ctx, UNetConfig, _UNetBase, _f32


@dataclass
class ControlNetOutput:
    down_block_res_samples: Tuple[torch.Tensor]
    mid_block_res_sample: torch.Tensor


class ControlNetConditioningEmbeddingSVD(nn.Module):
    """parameter holder + runner of the conditioning embedding (controlnet_sdv.py:64-119)"""

    def __init__(self, conditioning_embedding_channels: int, conditioning_channels: int = 3,
                 block_out_channels: Tuple[int, ...] = (16, 32, 96, 256)):
        super().__init__()
        if conditioning_channels > 8:
            raise LkgdHipError("conditioning images with more than 8 channels are not supported")
        if any(c % 16 for c in block_out_channels) or block_out_channels[-1] % 64:
            raise LkgdHipError("conditioning_embedding_out_channels must be multiples of 16 (the last one of 64)")
        self.conv_in = nn.Conv2d(conditioning_channels, block_out_channels[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(block_out_channels) - 1):
            cin, cout = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(nn.Conv2d(cin, cin, 3, padding=1))
            self.blocks.append(nn.Conv2d(cin, cout, 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(block_out_channels[-1], conditioning_embedding_channels, 3, padding=1)
        nn.init.zeros_(self.conv_out.weight)
        nn.init.zeros_(self.conv_out.bias)

    @staticmethod
    def _pack_small(conv: nn.Conv2d, cin_pad: int) -> Tuple[torch.Tensor, torch.Tensor]:
        w = conv.weight.detach().permute(0, 2, 3, 1)                       # [Cout, 3, 3, Cin]
        if w.shape[3] < cin_pad:
            w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[3]))
        return w.to(torch.float16).contiguous(), _f32(conv.bias)

    def pack(self):
        pk = SimpleNamespace()
        pk.first = self._pack_small(self.conv_in, 8)                        # 3 input channels padded to 8
        pk.blocks = [self._pack_small(b, b.in_channels) + (b.stride[0],) for b in self.blocks]
        pk.w_out, pk.b_out = pack_conv3x3(self.conv_out.weight.detach()), _f32(self.conv_out.bias)
        self._pk = pk

    def run_until_out(self, cond: torch.Tensor):
        """conditioning [B, F, C, H, W] -> (tokens [B*F*h*w, 256] before conv_out, h, w)"""
        if cond.dim() != 5:
            raise ValueError("controlnet_cond must be [batch, frames, channels, height, width]")
        b, f, c, H, W = cond.shape
        x = cond.to(dtype=torch.float16).reshape(b * f, c, H, W).contiguous()
        tok = torch.zeros(b * f * H * W, 8, dtype=torch.float16, device=x.device)    # channels-last, padded to 8 channels
        ops.nchw_to_tokens(x, out=tok[:, :c])
        pk = self._pk
        t = ops.conv3x3_small(tok, pk.first[0], pk.first[1], b * f, H, W, 1, True)
        for w, bias, stride in pk.blocks:
            t = ops.conv3x3_small(t, w, bias, b * f, H, W, stride, True)
            if stride == 2:
                H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        return t, H, W


class ControlNetSDVModel(_UNetBase):
    """reference models/controlnet_sdv.py ``ControlNetSDVModel``"""
    _encoder_only = True

    def __init__(self, config: Optional[UNetConfig] = None, conditioning_channels: int = 3,
                 conditioning_embedding_out_channels: Tuple[int, ...] = (16, 32, 96, 256), **kw):
        self._cn_args = (conditioning_channels, tuple(conditioning_embedding_out_channels))
        super().__init__(config, **kw)
        self.config.conditioning_channels = conditioning_channels
        self.config.conditioning_embedding_out_channels = tuple(conditioning_embedding_out_channels)
        boc = tuple(self.config.block_out_channels)
        n = len(boc)
        lpb = self.config.layers_per_block
        lpb = [lpb] * n if isinstance(lpb, int) else list(lpb)
        self.controlnet_cond_embedding = ControlNetConditioningEmbeddingSVD(boc[0], *self._cn_args)

        def zero_conv(ch):
            m = nn.Conv2d(ch, ch, 1)
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)
            return m
        self.controlnet_down_blocks = nn.ModuleList([zero_conv(boc[0])])
        for i in range(n):
            for _ in range(lpb[i]):
                self.controlnet_down_blocks.append(zero_conv(boc[i]))
            if i != n - 1:
                self.controlnet_down_blocks.append(zero_conv(boc[i]))
        self.controlnet_mid_block = zero_conv(boc[-1])
        self._cond_cache = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: Optional[str] = None, torch_dtype=None,
                        variant: Optional[str] = None, **_ignored):
        from .loading import build_from_pretrained
        return build_from_pretrained(cls, UNetConfig, pretrained_model_name_or_path, subfolder, torch_dtype, variant,
                                     conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256))

    def save_pretrained(self, save_directory: str, variant: Optional[str] = None, **_ignored):
        from .loading import save_pretrained
        save_pretrained(self, save_directory, dict(self.config.__dict__), type(self).__name__, variant)

    @classmethod
    def from_unet(cls, unet, controlnet_conditioning_channel_order: str = "rgb",
                  conditioning_embedding_out_channels: Tuple[int, ...] = (16, 32, 96, 256),
                  load_weights_from_unet: bool = True, conditioning_channels: int = 3):
        """controlnet_sdv.py:581-637: the UNet's config, and (optionally) its conv_in / embeddings / down blocks / mid block"""
        cfg = UNetConfig(**{k: v for k, v in unet.config.__dict__.items() if k in UNetConfig.__dataclass_fields__})
        c = cls(cfg, conditioning_channels=conditioning_channels,
                conditioning_embedding_out_channels=conditioning_embedding_out_channels)
        c = c.to(device=unet.device, dtype=unet.dtype)
        if load_weights_from_unet:
            for name in ("conv_in", "time_embedding", "add_embedding", "down_blocks", "mid_block"):
                getattr(c, name).load_state_dict(getattr(unet, name).state_dict())
            c.invalidate()
        return c

    def _pack_extra(self, pk):
        self.controlnet_cond_embedding.pack()
        pk.zero = [(pack_linear(m.weight.detach().reshape(m.out_channels, m.in_channels)), _f32(m.bias))
                   for m in self.controlnet_down_blocks]
        pk.zero_mid = (pack_linear(self.controlnet_mid_block.weight.detach().reshape(
            self.controlnet_mid_block.out_channels, -1)), _f32(self.controlnet_mid_block.bias))
        self._cond_cache = None

    def _cond_tokens(self, cond: torch.Tensor, B: int, F: int, H: int, W: int):
        """step-invariant part of the conditioning embedding (everything before conv_out).  One-entry cache that OWNS its
        key tensor (identity + in-place version): a new control video at a recycled address is never a hit."""
        c = self._cond_cache
        if c is not None and c[0] is cond and c[1] == cond._version:
            return c[2]
        t, h, w = self.controlnet_cond_embedding.run_until_out(cond.to(self.device))
        if (h, w) != (H, W) or t.shape[0] != B * F * H * W:
            raise ValueError(f"controlnet_cond of shape {tuple(cond.shape)} embeds to a {h}x{w} grid for "
                             f"{t.shape[0] // max(h * w, 1)} frames; the latents are {B * F} frames of {H}x{W}")
        self._cond_cache = (cond, cond._version, t)
        return t

    @torch.no_grad()
    def forward_tokens(self, tokens: torch.Tensor, B: int, F: int, H: int, W: int, timestep, encoder_hidden_states,
                       added_time_ids, controlnet_cond=None, conditioning_scale: float = 1.0, shard=None):
        """channels-last entry: input tokens [B*F*H*W, 8] -> (list of 12 residual token matrices, mid residual tokens).
        Under frame / CFG sharding (``shard``, lkgd_amd/dist_run.py) B, F, the tokens and ``controlnet_cond`` are the
        rank's LOCAL batch entries / frames; the encoder's temporal ops exchange exactly as the UNet's do."""
        from . import replay as _replay
        with _replay.invariant():
            self.prepare()
        ctx = Ctx(B, F, H, W, self.device, shard)
        pk = self._pk
        if getattr(pk, "has_lora", False):
            # the UNet builds a per-entry weight plan for wrapped projections (lkgd_amd/lora.py); this encoder does not, and
            # running the packed base weights would drop the adapters silently
            raise LkgdHipError("LoRA wrappers on the ControlNet are not applied by forward_tokens: lora.merge_lora() them first")
        self._time_embed(ctx, timestep, added_time_ids)
        self._cross_tables(ctx, encoder_hidden_states)
        c0 = pk.w_in.shape[0]
        h = ctx.new(ctx.T, c0)
        ops.gemm(tokens, pk.w_in, h, M=ctx.T, N=c0, K=128, bias=pk.b_in, mode=ops.A_CONV3X3_C8, Cin=8,
                 conv=(H, W, H, W, 1, 0))
        if controlnet_cond is not None:
            # sample = conv_in(sample) + conv_out(embedding): the add is the residual of conv_out's epilogue
            e = self._cond_tokens(controlnet_cond, B, F, H, W)
            ce = self.controlnet_cond_embedding._pk
            cin = e.shape[1]
            h2 = ctx.new(ctx.T, c0)
            ops.gemm(e, ce.w_out, h2, M=ctx.T, N=c0, K=9 * cin, bias=ce.b_out, mode=ops.A_CONV3X3, Cin=cin,
                     conv=(H, W, H, W, 1, 0), res1=h)
            h = h2
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk.run(ctx, h)
            skips += [o for o, _, _ in outs]
        h = self.mid_block.run(ctx, h)
        if len(skips) != len(pk.zero):
            raise LkgdHipError("internal: skip / zero-convolution count mismatch")
        down = []
        for s_, (w, b) in zip(skips, pk.zero):
            o = torch.empty_like(s_)
            ops.gemm(s_, w, o, M=s_.shape[0], N=w.shape[0], K=w.shape[1], bias=b, s_acc=float(conditioning_scale))
            down.append(o)
        mid = torch.empty_like(h)
        ops.gemm(h, pk.zero_mid[0], mid, M=h.shape[0], N=pk.zero_mid[0].shape[0], K=pk.zero_mid[0].shape[1],
                 bias=pk.zero_mid[1], s_acc=float(conditioning_scale))
        return down, mid, ctx

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, added_time_ids: torch.Tensor,
                controlnet_cond: Optional[torch.Tensor] = None, image_only_indicator: Optional[torch.Tensor] = None,
                return_dict: bool = True, guess_mode: bool = False, conditioning_scale: float = 1.0):
        if sample.dim() != 5:
            raise ValueError("sample must be [batch, frames, channels, height, width]")
        B, F, Cin, H, W = sample.shape
        x = sample.to(device=self.device, dtype=torch.float16).reshape(B * F, Cin, H, W).contiguous()
        down, mid, _ = self.forward_tokens(ops.nchw_to_tokens(x), B, F, H, W, timestep, encoder_hidden_states,
                                           added_time_ids, controlnet_cond, conditioning_scale)
        res: List[torch.Tensor] = []
        hh, ww, grid = H, W, {}
        for t in down + [mid]:
            # spatial size from the row count (levels halve with ceil-div, as the stride-2 convolutions do)
            hw = t.shape[0] // (B * F)
            while hh * ww != hw:
                hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
            res.append(ops.tokens_to_nchw(t, B * F, t.shape[1], hh, ww))
        down_n, mid_n = res[:-1], res[-1]
        if not return_dict:
            return (down_n, mid_n)
        return ControlNetOutput(down_block_res_samples=down_n, mid_block_res_sample=mid_n)
