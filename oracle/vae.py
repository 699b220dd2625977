"""Plain-torch CPU restatement of AutoencoderKL.encode (the VAE latent encode of the hot path).  TEST INFRASTRUCTURE.

Reference seam: VAECache.encode_images -> model.encode_with_vae -> vae.encode(samples).latent_dist.sample() -> scale_vae_latents_for_cache
(simpletuner/helpers/caching/vae.py:1238-1396, models/common.py:2767-2772, models/foundation_mixins.py:67-79: (z - shift_factor) * scaling_factor,
or z * scaling_factor when the VAE has no shift).  The network is diffusers' AutoencoderKL encoder (un-vendored; SURVEY.md Appendix A marks the
architecture UNCORROBORATED in-tree) restated from its published definition: conv_in 3x3 -> DownEncoderBlock2D x4 (ResnetBlock2D without time
embedding: GroupNorm32 eps 1e-6 -> SiLU -> conv3x3 -> GroupNorm -> SiLU -> conv3x3, 1x1 conv_shortcut on channel change; Downsample2D with
padding 0 = F.pad(x,(0,1,0,1)) + conv3x3 stride 2) -> UNetMidBlock2D (resnet, single-head attention of dim C with GroupNorm + residual, resnet)
-> GroupNorm -> SiLU -> conv_out (2*latent channels) [-> quant_conv 1x1] -> DiagonalGaussianDistribution(mean, logvar clamp [-30, 20]).
PINNED (round 3): the reference vendors this KL autoencoder once WITH a converter from diffusers' AutoencoderKL key names
(simpletuner/helpers/models/ideogram/autoencoder.py:29-273 network, :321-392 `convert_diffusers_state_dict`).  tools/gen_ref_models.py `gen_vae` executes it on a
seeded diffusers-named checkpoint (its converter accepted exactly this file's key names, its load_state_dict their shapes); tests/test_ref_models_cpu.py holds
`encode_moments` / `decode` and their input gradients to <= 1e-5 of its outputs (tests/golden/ref_vae_model.pt), SDXL-layout (quant convs) and FLUX-layout (none).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VAEConfig:
    # defaults = SDXL VAE (madebyollin/sdxl-vae-fp16-fix config): 4 latent channels, scaling 0.13025, quant_conv on
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.13025
    shift_factor: Optional[float] = None
    use_quant_conv: bool = True

    @staticmethod
    def flux():          # FLUX.1 VAE: 16 latent channels, shift + scale, no quant_conv
        return VAEConfig(latent_channels=16, scaling_factor=0.3611, shift_factor=0.1159, use_quant_conv=False)


def _conv(x, P, name, stride=1, padding=1):
    return F.conv2d(x, P[name + ".weight"], P.get(name + ".bias"), stride=stride, padding=padding)


def _resnet(P, p, x, groups):
    h = _conv(F.silu(F.group_norm(x, groups, P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-6)), P, p + "conv1")
    h = _conv(F.silu(F.group_norm(h, groups, P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-6)), P, p + "conv2")
    if (p + "conv_shortcut.weight") in P:
        x = _conv(x, P, p + "conv_shortcut", padding=0)
    return x + h


def _mid_attention(P, p, x, groups):
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, P[p + "group_norm.weight"], P[p + "group_norm.bias"], 1e-6).view(B, C, H * W).transpose(1, 2)
    q = F.linear(h, P[p + "to_q.weight"], P[p + "to_q.bias"])
    k = F.linear(h, P[p + "to_k.weight"], P[p + "to_k.bias"])
    v = F.linear(h, P[p + "to_v.weight"], P[p + "to_v.bias"])
    o = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1) @ v
    o = F.linear(o, P[p + "to_out.0.weight"], P[p + "to_out.0.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def encode_moments(P: Dict[str, torch.Tensor], cfg: VAEConfig, x: torch.Tensor):
    """AutoencoderKL.encode -> the distribution parameters [B, 2*latent, H/8, W/8] (mean | logvar)"""
    g = cfg.norm_num_groups
    h = _conv(x, P, "encoder.conv_in")
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = _resnet(P, f"encoder.down_blocks.{i}.resnets.{j}.", h, g)
        if i < nb - 1:
            h = _conv(F.pad(h, (0, 1, 0, 1)), P, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
    h = _resnet(P, "encoder.mid_block.resnets.0.", h, g)
    h = _mid_attention(P, "encoder.mid_block.attentions.0.", h, g)
    h = _resnet(P, "encoder.mid_block.resnets.1.", h, g)
    h = F.silu(F.group_norm(h, g, P["encoder.conv_norm_out.weight"], P["encoder.conv_norm_out.bias"], 1e-6))
    h = _conv(h, P, "encoder.conv_out")
    if cfg.use_quant_conv:
        h = _conv(h, P, "quant_conv", padding=0)
    return h


def sample_and_scale(moments: torch.Tensor, cfg: VAEConfig, eps: Optional[torch.Tensor] = None):
    """DiagonalGaussianDistribution.sample (mean + exp(0.5*clamp(logvar,-30,20)) * eps; eps None -> mode) then scale_vae_latents_for_cache"""
    mean, logvar = moments.chunk(2, dim=1)
    z = mean if eps is None else mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * eps
    if cfg.shift_factor is not None:
        return (z - cfg.shift_factor) * cfg.scaling_factor
    return z * cfg.scaling_factor


from tools.flop_count import vae_encoder_flops as encoder_flops  # noqa: E402,F401


# ---- decoder (validation images; the product decoder runs on the same conv / GroupNorm /
# upsample kernels).  diffusers' Decoder: conv_in (latent -> C_last) -> UNetMidBlock2D -> UpDecoderBlock2D x4 over the REVERSED channel list
# (layers_per_block + 1 resnets each, nearest-2x Upsample2D + conv3x3 on all but the last) -> GroupNorm -> SiLU -> conv_out (C_0 -> 3);
# AutoencoderKL.decode applies post_quant_conv (1x1 on the latents) first when the VAE has one.  PINNED like the encoder (the vendored Decoder,
# simpletuner/helpers/models/ideogram/autoencoder.py:189-273, executed through its converter: tests/test_ref_models_cpu.py); the
# parameter totals of init_params (encoder + decoder + quant convs) equal the published SD / SDXL VAE size, 83,653,863.
def unscale_latents(z: torch.Tensor, cfg: VAEConfig) -> torch.Tensor:
    """inverse of scale_vae_latents_for_cache: what the pipelines do before vae.decode (z / scaling_factor + shift_factor)"""
    z = z / cfg.scaling_factor
    return z + cfg.shift_factor if cfg.shift_factor is not None else z


def decode(P: Dict[str, torch.Tensor], cfg: VAEConfig, z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.decode(z).sample -> [B, 3, 8H, 8W]"""
    g = cfg.norm_num_groups
    if cfg.use_quant_conv:
        z = _conv(z, P, "post_quant_conv", padding=0)
    h = _conv(z, P, "decoder.conv_in")
    h = _resnet(P, "decoder.mid_block.resnets.0.", h, g)
    h = _mid_attention(P, "decoder.mid_block.attentions.0.", h, g)
    h = _resnet(P, "decoder.mid_block.resnets.1.", h, g)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(P, f"decoder.up_blocks.{i}.resnets.{j}.", h, g)
        if i < nb - 1:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), P, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    h = F.silu(F.group_norm(h, g, P["decoder.conv_norm_out.weight"], P["decoder.conv_norm_out.bias"], 1e-6))
    return _conv(h, P, "decoder.conv_out")


def init_params(cfg: VAEConfig, seed: int = 0, shapes_only: bool = False) -> Dict[str, torch.Tensor]:
    """random weights under diffusers' AutoencoderKL names (encoder, decoder, quant / post_quant convs), fan-in scaled"""
    gen = torch.Generator().manual_seed(seed)
    P: Dict[str, torch.Tensor] = {}

    def rnd(*shape):
        return torch.empty(*shape, device="meta") if shapes_only else torch.randn(*shape, generator=gen)

    def conv(name, ci, co, k=3):
        P[name + ".weight"] = rnd(co, ci, k, k) * (1.0 / math.sqrt(ci * k * k))
        P[name + ".bias"] = 0.02 * rnd(co)

    def lin(name, ci, co):
        P[name + ".weight"] = rnd(co, ci) * (1.0 / math.sqrt(ci))
        P[name + ".bias"] = 0.02 * rnd(co)

    def norm(name, c):
        P[name + ".weight"] = 1.0 + 0.1 * rnd(c)
        P[name + ".bias"] = 0.02 * rnd(c)

    def res(p, ci, co):
        norm(p + "norm1", ci); conv(p + "conv1", ci, co); norm(p + "norm2", co); conv(p + "conv2", co, co)
        if ci != co:
            conv(p + "conv_shortcut", ci, co, 1)

    def mid(p, c):
        res(p + "resnets.0.", c, c)
        a = p + "attentions.0."
        norm(a + "group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(a + n, c, c)
        res(p + "resnets.1.", c, c)

    ch, L = cfg.block_out_channels, cfg.latent_channels
    conv("encoder.conv_in", cfg.in_channels, ch[0])
    cin = ch[0]
    for i, co in enumerate(ch):
        for j in range(cfg.layers_per_block):
            res(f"encoder.down_blocks.{i}.resnets.{j}.", cin, co)
            cin = co
        if i < len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cin, cin)
    mid("encoder.mid_block.", cin)
    norm("encoder.conv_norm_out", cin); conv("encoder.conv_out", cin, 2 * L)
    if cfg.use_quant_conv:
        conv("quant_conv", 2 * L, 2 * L, 1); conv("post_quant_conv", L, L, 1)
    rev = tuple(reversed(ch))
    conv("decoder.conv_in", L, rev[0])
    mid("decoder.mid_block.", rev[0])
    cin = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}.", cin, co)
            cin = co
        if i < len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cin, cin)
    norm("decoder.conv_norm_out", cin); conv("decoder.conv_out", cin, cfg.in_channels)
    return P
