"""AutoencoderKL — the VAE latent encode of the hot path on MI355X, and the decode of the validation loop.

Mirrors what the reference drives (caching/vae.py:1238-1396 -> models/common.py:2767-2772): `vae.encode(samples)` returns an object whose
`.latent_dist` has `.sample(generator=None)`, `.mode()`, `.parameters`; `vae.config.scaling_factor / shift_factor` feed
`scale_vae_latents_for_cache` (foundation_mixins.py:67-79), provided here as `encode_scaled()` too (one fused pass).  State-dict keys are diffusers'
(`encoder.conv_in.weight` ..., conv weights [O,I,kh,kw]); the native storage is [O, 9*I] per conv.  Everything is a libst355 launch:
grid-buffer convolutions-as-GEMM (no im2col except conv_in's 3 -> 8 channels and the three stride-2 (0,1,0,1)-padded downsample convs),
GroupNorm+SiLU kernels, and the single-head dim-512 mid-block attention as two plain GEMMs around a row-softmax kernel (the score matrix
[HW, HW] bf16 is materialised per image: 512 MiB at 1024^2 — nothing on a 288 GB part).  quant_conv (1x1 on 2L channels) is folded into
conv_out at load time (both are linear).  Inference only (the VAE is frozen during training).

Decoder (`decode`, `decode_scaled`): diffusers' Decoder as the validation pipelines call it (`vae.decode(z / scaling_factor + shift_factor).sample`,
e.g. flux/pipeline.py) — [post_quant_conv 1x1] -> conv_in -> UNetMidBlock2D -> 4 UpDecoderBlock2D over the reversed channel list
(layers_per_block + 1 resnets, nearest-2x upsample + conv3x3 on all but the last) -> GroupNorm -> SiLU -> conv_out — on the same grid-buffer
convolution / GroupNorm / upsample kernels.  The latent (4 / 16 channels) enters as a 64-channel grid (zero channels above L: the convolution
GEMM's K granule), conv_out's 3 channels leave through an 8-wide output.  Loaded only when the state dict carries `decoder.*` keys.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import ops

BF16 = torch.bfloat16
F32 = torch.float32


class DiagonalGaussian:
    """diffusers DiagonalGaussianDistribution over the encoder's [B, 2L, h, w] moments (logvar clamped to [-30, 20])"""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, logvar = parameters.float().chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=F32)
        return (self.mean + self.std * eps).to(self.parameters.dtype)

    def mode(self):
        return self.mean.to(self.parameters.dtype)


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels: int = 3, latent_channels: int = 4, block_out_channels=(128, 256, 512, 512), layers_per_block: int = 2,
                 norm_num_groups: int = 32, scaling_factor: float = 0.13025, shift_factor: Optional[float] = None, use_quant_conv: bool = True,
                 device=None, **_ignored):
        super().__init__()
        if in_channels > 8 or any(c % 64 for c in block_out_channels) or (2 * latent_channels) % 8:
            raise ValueError("AutoencoderKL(st355): in_channels <= 8, block_out_channels multiples of 64, 2*latent_channels a multiple of 8")
        self.config = SimpleNamespace(in_channels=in_channels, latent_channels=latent_channels, block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, norm_num_groups=norm_num_groups, scaling_factor=scaling_factor,
                                      shift_factor=shift_factor, use_quant_conv=use_quant_conv)
        self.device_ = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dtype = BF16
        self.W: Dict[str, torch.Tensor] = {}          # native-layout weights (bf16, device)

    @property
    def device(self):
        return self.device_

    # ---- weights ----
    def _names(self):
        c = self.config
        ch = c.block_out_channels
        names = [("encoder.conv_in", "conv", c.in_channels, ch[0], 3)]
        cin = ch[0]

        def res(p, ci, co):
            out = [(p + "norm1", "norm", ci, ci, 0), (p + "conv1", "conv", ci, co, 3), (p + "norm2", "norm", co, co, 0), (p + "conv2", "conv", co, co, 3)]
            if ci != co:
                out.append((p + "conv_shortcut", "conv", ci, co, 1))
            return out

        for i, co in enumerate(ch):
            for j in range(c.layers_per_block):
                names += res(f"encoder.down_blocks.{i}.resnets.{j}.", cin, co)
                cin = co
            if i < len(ch) - 1:
                names.append((f"encoder.down_blocks.{i}.downsamplers.0.conv", "conv", cin, cin, 3))
        names += res("encoder.mid_block.resnets.0.", cin, cin)
        a = "encoder.mid_block.attentions.0."
        names += [(a + "group_norm", "norm", cin, cin, 0)] + [(a + n, "lin", cin, cin, 0) for n in ("to_q", "to_k", "to_v", "to_out.0")]
        names += res("encoder.mid_block.resnets.1.", cin, cin)
        names += [("encoder.conv_norm_out", "norm", cin, cin, 0), ("encoder.conv_out", "conv", cin, 2 * c.latent_channels, 3)]
        if c.use_quant_conv:
            names.append(("quant_conv", "conv", 2 * c.latent_channels, 2 * c.latent_channels, 1))
        return names

    def _decoder_names(self):
        c = self.config
        rev = tuple(reversed(c.block_out_channels))
        L = c.latent_channels

        def res(p, ci, co):
            out = [(p + "norm1", "norm", ci, ci, 0), (p + "conv1", "conv", ci, co, 3), (p + "norm2", "norm", co, co, 0), (p + "conv2", "conv", co, co, 3)]
            if ci != co:
                out.append((p + "conv_shortcut", "conv", ci, co, 1))
            return out

        names = []
        if c.use_quant_conv:
            names.append(("post_quant_conv", "conv", L, L, 1))
        names.append(("decoder.conv_in", "conv", L, rev[0], 3))
        names += res("decoder.mid_block.resnets.0.", rev[0], rev[0])
        a = "decoder.mid_block.attentions.0."
        names += [(a + "group_norm", "norm", rev[0], rev[0], 0)] + [(a + n, "lin", rev[0], rev[0], 0) for n in ("to_q", "to_k", "to_v", "to_out.0")]
        names += res("decoder.mid_block.resnets.1.", rev[0], rev[0])
        cin = rev[0]
        for i, co in enumerate(rev):
            for j in range(c.layers_per_block + 1):
                names += res(f"decoder.up_blocks.{i}.resnets.{j}.", cin, co)
                cin = co
            if i < len(rev) - 1:
                names.append((f"decoder.up_blocks.{i}.upsamplers.0.conv", "conv", cin, cin, 3))
        names += [("decoder.conv_norm_out", "norm", cin, cin, 0), ("decoder.conv_out", "conv", cin, c.in_channels, 3)]
        return names

    @torch.no_grad()
    def synthetic_state_dict(self, seed: int = 0, decoder: bool = False) -> Dict[str, torch.Tensor]:
        """random weights in diffusers' names / shapes (bf16-representable fp32), for parity tests and benches without a checkpoint"""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, kind, ci, co, k in self._names() + (self._decoder_names() if decoder else []):
            if kind == "norm":
                sd[name + ".weight"] = (1.0 + 0.1 * torch.randn(ci, generator=g)).to(BF16).float()
                sd[name + ".bias"] = (0.02 * torch.randn(ci, generator=g)).to(BF16).float()
            elif kind == "lin":
                sd[name + ".weight"] = (torch.randn(co, ci, generator=g) / math.sqrt(ci)).to(BF16).float()
                sd[name + ".bias"] = (0.02 * torch.randn(co, generator=g)).to(BF16).float()
            else:
                sd[name + ".weight"] = (torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)).to(BF16).float()
                sd[name + ".bias"] = (0.02 * torch.randn(co, generator=g)).to(BF16).float()
        return sd

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):   # noqa: D401 - diffusers-layout state dict
        dev, c = self.device_, self.config
        W = {}
        for name, kind, ci, co, k in self._names():
            w, b = sd[name + ".weight"].float(), sd[name + ".bias"].float()
            if kind == "norm" or kind == "lin":
                W[name + ".weight"], W[name + ".bias"] = w.to(dev, BF16).contiguous(), b.to(dev, BF16).contiguous()
            elif name == "encoder.conv_in":
                w8 = torch.zeros(co, 128)
                w8[:, :72].view(co, 9, 8)[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
                W[name + ".weight"], W[name + ".bias"] = w8.to(dev, BF16), b.to(dev, BF16)
            elif name == "quant_conv":
                continue
            else:
                if name == "encoder.conv_out" and c.use_quant_conv:     # fold the 1x1 quant_conv: W' = Wq Wo, b' = Wq bo + bq  (fp32, one rounding)
                    wq, bq = sd["quant_conv.weight"].float().view(co, co), sd["quant_conv.bias"].float()
                    w = torch.einsum("qo,oikl->qikl", wq, w)
                    b = wq @ b + bq
                W[name + ".weight"] = w.permute(0, 2, 3, 1).reshape(co, -1).to(dev, BF16).contiguous()
                W[name + ".bias"] = b.to(dev, BF16).contiguous()
        self.has_decoder = "decoder.conv_in.weight" in sd
        if self.has_decoder:
            L = c.latent_channels
            for name, kind, ci, co, k in self._decoder_names():
                w, b = sd[name + ".weight"].float(), sd[name + ".bias"].float()
                if kind == "norm" or kind == "lin":
                    W[name + ".weight"], W[name + ".bias"] = w.to(dev, BF16).contiguous(), b.to(dev, BF16).contiguous()
                elif name == "post_quant_conv":                    # 1x1 on the latents: [64 (L used), 64 (L used)] over the 64-channel latent grid
                    w64, b64 = torch.zeros(64, 64), torch.zeros(64)
                    w64[:L, :L], b64[:L] = w.view(L, L), b
                    W[name + ".weight"], W[name + ".bias"] = w64.to(dev, BF16), b64.to(dev, BF16)
                elif name == "decoder.conv_in":                    # reads the 64-channel latent grid: zero weights on the padding channels
                    w64 = torch.zeros(co, 9, 64)
                    w64[:, :, :L] = w.permute(0, 2, 3, 1).reshape(co, 9, L)
                    W[name + ".weight"], W[name + ".bias"] = w64.reshape(co, 9 * 64).to(dev, BF16).contiguous(), b.to(dev, BF16)
                elif name == "decoder.conv_out":                   # 3 output channels through an 8-wide output (Cout granule of the conv GEMM)
                    w8, b8 = torch.zeros(8, 9 * ci), torch.zeros(8)
                    w8[:co], b8[:co] = w.permute(0, 2, 3, 1).reshape(co, -1), b
                    W[name + ".weight"], W[name + ".bias"] = w8.to(dev, BF16).contiguous(), b8.to(dev, BF16)
                else:
                    W[name + ".weight"] = w.permute(0, 2, 3, 1).reshape(co, -1).to(dev, BF16).contiguous()
                    W[name + ".bias"] = b.to(dev, BF16).contiguous()
        for a in ("encoder.mid_block.attentions.0.",) + (("decoder.mid_block.attentions.0.",) if self.has_decoder else ()):
            W[a + "qkv.weight"] = torch.cat([W[a + n + ".weight"] for n in ("to_q", "to_k", "to_v")], 0).contiguous()
            W[a + "qkv.bias"] = torch.cat([W[a + n + ".bias"] for n in ("to_q", "to_k", "to_v")], 0).contiguous()
        self.W = W
        return self

    # ---- forward ----
    def _res(self, p, x, B, H, Wd):
        W = self.W
        g = self.config.norm_num_groups
        h, _ = ops.groupnorm_fwd(x, W[p + "norm1.weight"], W[p + "norm1.bias"], B, H, Wd, groups=g, eps=1e-6, silu=True)
        h = ops.conv(h, W[p + "conv1.weight"], B, H, Wd, bias=W[p + "conv1.bias"])
        h, _ = ops.groupnorm_fwd(h, W[p + "norm2.weight"], W[p + "norm2.bias"], B, H, Wd, groups=g, eps=1e-6, silu=True)
        sc = x
        if (p + "conv_shortcut.weight") in W:
            sc = ops.conv(x, W[p + "conv_shortcut.weight"], B, H, Wd, bias=W[p + "conv_shortcut.bias"], taps=1)
        return ops.conv(h, W[p + "conv2.weight"], B, H, Wd, bias=W[p + "conv2.bias"], residual=sc)

    @torch.no_grad()
    def encode_moments(self, x: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] pixels (any float dtype) -> [B, 2L, H/8, W/8] bf16 distribution parameters (mean | logvar)"""
        if not self.W:
            raise RuntimeError("AutoencoderKL(st355): load_state_dict() first")
        c, W = self.config, self.W
        B, _, H, Wd = x.shape
        nb = len(c.block_out_channels)
        if H % (1 << (nb - 1)) or Wd % (1 << (nb - 1)):
            raise ValueError("image sides must be divisible by 2^(levels-1)")
        # ONE C entry point (st355_vae_encode): the whole encoder — conv_in over pre-gathered columns, the DownEncoderBlock2D levels, the mid
        # block with its single-head attention, GroupNorm + SiLU + conv_out — is sequenced inside libst355 over a caller-owned workspace
        return ops.vae_encode(self._encoder_table(), x.to(device=self.device_, dtype=BF16))

    def _encoder_table(self):
        """the device-pointer table of st355_vae_encode, in the walk include/st355.h states (conv weights in the native [Cout, taps*Cin] layout)"""
        t = getattr(self, "_enc_table", None)
        if t is None or t[0] is not self.W:
            c, W = self.config, self.W
            ts = []
            a = "encoder.mid_block.attentions.0."
            for name, kind, ci, co, k in self._names():
                if name == "quant_conv" or (name.startswith(a) and name[len(a):] in ("to_k", "to_v")):
                    continue                              # folded into conv_out / stacked into the qkv matrix by load_state_dict
                key = a + "qkv" if name == a + "to_q" else name
                ts += [W[key + ".weight"], W[key + ".bias"]]
            t = (self.W, ops.VaeEncoderTable(c.in_channels, c.latent_channels, c.block_out_channels, c.layers_per_block, c.norm_num_groups, ts))
            self._enc_table = t
        return t[1]

    def _mid_attention(self, x, B, H, Wd, a):
        W = self.W
        C_ = x.shape[1]
        S = H * Wd
        n, _ = ops.groupnorm_fwd(x, W[a + "group_norm.weight"], W[a + "group_norm.bias"], B, H, Wd, groups=self.config.norm_num_groups, eps=1e-6, silu=False,
                                 out_tokens=True)
        qkv = ops.gemm(n, W[a + "qkv.weight"], bias=W[a + "qkv.bias"])                     # [B*S, 3C]
        o = torch.empty(B * S, C_, dtype=BF16, device=x.device)
        Sp = (S + 63) // 64 * 64
        for b in range(B):                                                                  # one head of dim C: scores are a plain [S, S] GEMM per image
            q, k, v = (qkv[b * S:(b + 1) * S, i * C_:(i + 1) * C_] for i in range(3))
            scores = ops.gemm(q, k)                                                         # q k^T  [S, S] bf16
            ops.softmax_rows_(scores, 1.0 / math.sqrt(C_))
            if Sp != S:                                                                     # contraction granule 64: zero-padded probabilities / V^T
                pp = torch.zeros(S, Sp, dtype=BF16, device=x.device); pp[:, :S] = scores; scores = pp
            vt = torch.zeros(C_, Sp, dtype=BF16, device=x.device)
            vt[:, :S] = v.t()
            ops.gemm(scores, vt, out=o[b * S:(b + 1) * S])
        out = ops.gemm(o, W[a + "to_out.0.weight"], bias=W[a + "to_out.0.bias"])
        return ops.tokens_to_grid(out, B, H, Wd, residual=x)

    @torch.no_grad()
    def encode(self, x, return_dict: bool = True):
        dist = DiagonalGaussian(self.encode_moments(x))
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def encode_scaled(self, x, generator=None, sample: bool = True):
        """vae.encode(x).latent_dist.sample() then scale_vae_latents_for_cache (foundation_mixins.py:67-79)"""
        d = DiagonalGaussian(self.encode_moments(x))
        z = d.sample(generator) if sample else d.mode()
        c = self.config
        if c.shift_factor is not None:
            return ((z.float() - c.shift_factor) * c.scaling_factor).to(z.dtype)
        return (z.float() * c.scaling_factor).to(z.dtype)


    # ---- decode (validation images) ----
    @torch.no_grad()
    def decode(self, z, return_dict: bool = True):
        """AutoencoderKL.decode(z).sample: [B, L, h, w] latents (already un-scaled) -> [B, 3, 8h, 8w] bf16"""
        if not getattr(self, "has_decoder", False):
            raise RuntimeError("AutoencoderKL(st355): the loaded state dict carries no decoder.* weights")
        c, W = self.config, self.W
        B, L, H, Wd = z.shape
        if L != c.latent_channels:
            raise ValueError(f"decode: expected {c.latent_channels} latent channels, got {L}")
        h = ops.grid_from_nchw(z.to(device=self.device_, dtype=BF16), 64)
        if c.use_quant_conv:
            h = ops.conv(h, W["post_quant_conv.weight"], B, H, Wd, bias=W["post_quant_conv.bias"], taps=1)
        h = ops.conv(h, W["decoder.conv_in.weight"], B, H, Wd, bias=W["decoder.conv_in.bias"])
        h = self._res("decoder.mid_block.resnets.0.", h, B, H, Wd)
        h = self._mid_attention(h, B, H, Wd, "decoder.mid_block.attentions.0.")
        h = self._res("decoder.mid_block.resnets.1.", h, B, H, Wd)
        nb = len(c.block_out_channels)
        for i in range(nb):
            for j in range(c.layers_per_block + 1):
                h = self._res(f"decoder.up_blocks.{i}.resnets.{j}.", h, B, H, Wd)
            if i < nb - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                h = ops.upsample2x(h, B, H, Wd)               # Upsample2D: nearest 2x, then conv3x3
                H, Wd = 2 * H, 2 * Wd
                h = ops.conv(h, W[p + ".weight"], B, H, Wd, bias=W[p + ".bias"])
        h, _ = ops.groupnorm_fwd(h, W["decoder.conv_norm_out.weight"], W["decoder.conv_norm_out.bias"], B, H, Wd, groups=c.norm_num_groups, eps=1e-6, silu=True)
        y = ops.conv(h, W["decoder.conv_out.weight"], B, H, Wd, bias=W["decoder.conv_out.bias"])
        img = ops.grid_to_nchw(y, B, c.in_channels, H, Wd)
        return SimpleNamespace(sample=img) if return_dict else (img,)

    @torch.no_grad()
    def decode_scaled(self, z):
        """what the validation pipelines do with cache-scaled latents: vae.decode(z / scaling_factor + shift_factor).sample"""
        c = self.config
        z = z.float() / c.scaling_factor
        if c.shift_factor is not None:
            z = z + c.shift_factor
        return self.decode(z.to(BF16), return_dict=False)[0]
