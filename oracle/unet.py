"""Plain-torch CPU restatement of the conv UNet forward of SDXL / SD1.5 (autograd supplies the backward).  TEST INFRASTRUCTURE.

The reference calls the network positionally and never vendors it:
    SDXL._model_predict_single        simpletuner/helpers/models/sdxl/model.py:306-373  (sample, timestep, encoder_hidden_states,
                                      added_cond_kwargs={"text_embeds", "time_ids"}, return_dict=False)[0]
    StableDiffusion1._model_predict_single   simpletuner/helpers/models/sd1x/model.py:224-270
    FlowMapUNet2DConditionModel       simpletuner/helpers/models/unet_flowmap.py:23-44 (a thin subclass of diffusers' class)
The arithmetic is diffusers' `UNet2DConditionModel` (>= 0.36, un-vendored) — restated here from its published
module definitions: Timesteps(flip_sin_to_cos, shift 0) -> TimestepEmbedding; "text_time" addition embedding (SDXL); conv_in; Down /
CrossAttnDown blocks of ResnetBlock2D (GroupNorm32 eps 1e-5 -> SiLU -> conv3x3 -> + time_emb_proj(SiLU(emb)) -> GroupNorm -> SiLU -> conv3x3,
1x1 conv_shortcut when channels change) and Transformer2DModel (GroupNorm32 eps 1e-6 -> proj_in -> BasicTransformerBlock x N: LayerNorm ->
self-attention -> LayerNorm -> cross-attention over the text tokens -> LayerNorm -> GEGLU feed-forward, all residual -> proj_out -> + input);
Downsample2D (conv3x3 stride 2 pad 1); mid block; Up blocks (concat skip, resnets, nearest-2x Upsample2D + conv3x3); conv_norm_out -> SiLU ->
conv_out.  State-dict keys and tensor shapes are diffusers' (conv weights [O,I,kh,kw]).
PARITY: LEAVES PINNED, DOWN HALF OF THE WALK PINNED, BLOCK INTERIORS AND THE UP PATH RESTATED.  The reference holds no golden tensor for this network
 and diffusers is not installable here, but (a) `unet_down_mid` — the embeddings incl. SDXL's "text_time" embedder, conv_in, the order of the skip
tensors, the mid block, the head-count meaning of `attention_head_dim` — is checked against diffusers' ControlNetModel, which the reference VENDORS
(helpers/models/kolors/controlnet.py:132-931: the same constructor wiring and forward up to the mid block), executed in the build container over block stand-ins made
of this file's leaves (tools/gen_ref_unet_walk.py -> tests/golden/ref_unet_walk.pt, tests/test_ref_unet_walk_cpu.py: bit-identical outputs and gradients), and (b)
the leaves below are checked against reference code executed in the build container (tools/gen_ref_unet_leaves.py -> tests/golden/ref_unet_leaves.pt,
tests/test_ref_unet_leaves_cpu.py, forward and every gradient <= 1e-5):
  * `resnet` (norm -> SiLU -> conv3x3 -> norm -> SiLU -> conv3x3, GroupNorm(32), 1x1 conv_shortcut), `upsample` (nearest 2x + conv3x3), the stride-2 3x3
    convolution of `downsample`, and `_attention` + GroupNorm + residual against the KL-autoencoder blocks the reference vendors
    (helpers/models/ideogram/autoencoder.py:29-110) fed through its own diffusers-key converter (:321-392);
  * `timestep_proj` + the Linear -> SiLU -> Linear embedder against Timesteps / TimestepEmbedding lifted from helpers/models/heartmula/codec/transformer.py.
What stays RESTATED (from the published diffusers modules; the reference carries no copy): the `+ time_emb_proj(SiLU(emb))` add inside `resnet` (one line),
the symmetric padding-1 of Downsample2D, `basic_block` / GEGLU / multi-head cross-attention (cross-checked only against tools/ref_shim.py's independent
restatement of the same public definition), the GroupNorm(1e-6) -> proj_in -> blocks -> proj_out wiring of `transformer2d`, the order resnet -> attention ->
(downsampler) inside a block, and the second half of `unet_forward` (the reversed up path with skip concatenation, conv_norm_out, conv_out).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .flux import timestep_proj


@dataclass
class UNetConfig:
    # defaults = stabilityai/stable-diffusion-xl-base-1.0 unet/config.json
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    layers_per_block: int = 2
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    attention_head_dim: Tuple[int, ...] = (5, 10, 20)           # (mis-named in diffusers: the NUMBER of heads per block)
    cross_attention_dim: int = 2048
    use_linear_projection: bool = True
    addition_embed_type: Optional[str] = "text_time"
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    norm_num_groups: int = 32
    norm_eps: float = 1e-5

    @staticmethod
    def sd15():
        return UNetConfig(block_out_channels=(320, 640, 1280, 1280),
                          down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                          up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                          transformer_layers_per_block=(1, 1, 1, 1), attention_head_dim=(8, 8, 8, 8), cross_attention_dim=768,
                          use_linear_projection=False, addition_embed_type=None)


def _lin(x, P, name):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _conv(x, P, name, stride=1, padding=1):
    return F.conv2d(x, P[name + ".weight"], P.get(name + ".bias"), stride=stride, padding=padding)


def resnet(P, p, x, emb, groups, eps):
    h = F.silu(F.group_norm(x, groups, P[p + "norm1.weight"], P[p + "norm1.bias"], eps))
    h = _conv(h, P, p + "conv1")
    h = h + _lin(F.silu(emb), P, p + "time_emb_proj")[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, P[p + "norm2.weight"], P[p + "norm2.bias"], eps))
    h = _conv(h, P, p + "conv2")
    if (p + "conv_shortcut.weight") in P:
        x = _conv(x, P, p + "conv_shortcut", padding=0)
    return x + h


def downsample(P, name, x, padding=1):
    """Downsample2D: 3x3 convolution, stride 2 (diffusers' UNet form: symmetric padding 1)"""
    return _conv(x, P, name, stride=2, padding=padding)


def upsample(P, name, x):
    """Upsample2D: nearest-neighbour 2x, then 3x3 convolution"""
    return _conv(F.interpolate(x, scale_factor=2.0, mode="nearest"), P, name)


def time_embedding(P, name, t_emb):
    """TimestepEmbedding: Linear -> SiLU -> Linear"""
    return _lin(F.silu(_lin(t_emb, P, name + ".linear_1")), P, name + ".linear_2")


def _attention(P, p, x, ctx, heads):
    B, S, C = x.shape
    q = _lin(x, P, p + "to_q")
    k = _lin(ctx, P, p + "to_k")
    v = _lin(ctx, P, p + "to_v")
    d = C // heads
    q, k, v = (t.view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    o = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v
    return _lin(o.transpose(1, 2).reshape(B, S, C), P, p + "to_out.0")


def basic_block(P, p, h, ctx, heads):
    C = h.shape[-1]
    n = F.layer_norm(h, (C,), P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-5)
    h = _attention(P, p + "attn1.", n, n, heads) + h
    n = F.layer_norm(h, (C,), P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-5)
    h = _attention(P, p + "attn2.", n, ctx, heads) + h
    n = F.layer_norm(h, (C,), P[p + "norm3.weight"], P[p + "norm3.bias"], 1e-5)
    val, gate = _lin(n, P, p + "ff.net.0.proj").chunk(2, dim=-1)
    return _lin(val * F.gelu(gate), P, p + "ff.net.2") + h


def transformer2d(P, p, x, ctx, heads, n_layers, groups, linear_proj):
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, P[p + "norm.weight"], P[p + "norm.bias"], 1e-6)
    if linear_proj:
        h = _lin(h.permute(0, 2, 3, 1).reshape(B, H * W, C), P, p + "proj_in")
    else:
        h = _conv(h, P, p + "proj_in", padding=0).permute(0, 2, 3, 1).reshape(B, H * W, C)
    for k in range(n_layers):
        h = basic_block(P, f"{p}transformer_blocks.{k}.", h, ctx, heads)
    if linear_proj:
        h = _lin(h, P, p + "proj_out").reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = _conv(h.reshape(B, H, W, C).permute(0, 3, 1, 2), P, p + "proj_out", padding=0)
    return h + res


def unet_down_mid(P: Dict[str, torch.Tensor], cfg: UNetConfig, sample, timesteps, encoder_hidden_states, added_cond_kwargs=None, cond_residual=None):
    """The first half of UNet2DConditionModel.forward: embeddings -> conv_in -> down blocks (every resnet / attention / downsampler output is a skip) -> mid
    block.  Returns (skips, mid, emb).  PINNED: this is also the whole of diffusers' ControlNetModel.forward up to its 1x1 output convolutions, and the reference
    vendors that class (simpletuner/helpers/models/kolors/controlnet.py:132-931: its constructor wiring — time_embed_dim = 4 * block_out_channels[0], the
    "text_time" embedder over projection_class_embeddings_input_dim, `num_attention_heads or attention_head_dim` as the HEAD COUNT — and its forward:
    time_ids.flatten() -> add_time_proj -> reshape(B, -1) -> cat([text_embeds, time_embeds]) -> add_embedding, emb = time + aug, conv_in (+ the conditioning
    embedding: `cond_residual`), the down-block loop collecting (sample,) + res_samples, the mid block).  tools/gen_ref_unet_walk.py executes it over block
    stand-ins built from this file's leaves; tests/test_ref_unet_walk_cpu.py holds this function to its outputs and gradients <= 1e-5."""
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    dt = sample.dtype
    c0 = cfg.block_out_channels[0]
    t_emb = timestep_proj(timesteps.to(sample.device).expand(sample.shape[0]), c0).to(dt)
    emb = time_embedding(P, "time_embedding", t_emb)
    if cfg.addition_embed_type == "text_time":
        text_embeds, time_ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
        tid = timestep_proj(time_ids.flatten(), cfg.addition_time_embed_dim).reshape(text_embeds.shape[0], -1)
        add = torch.cat([text_embeds.to(dt), tid.to(dt)], dim=-1)
        emb = emb + time_embedding(P, "add_embedding", add)
    ctx = encoder_hidden_states.to(dt)
    x = _conv(sample, P, "conv_in")
    if cond_residual is not None:
        x = x + cond_residual
    skips = [x]
    nb = len(cfg.block_out_channels)
    for i, typ in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            x = resnet(P, f"down_blocks.{i}.resnets.{j}.", x, emb, g, eps)
            if typ.startswith("CrossAttn"):
                x = transformer2d(P, f"down_blocks.{i}.attentions.{j}.", x, ctx, cfg.attention_head_dim[i], cfg.transformer_layers_per_block[i], g,
                                  cfg.use_linear_projection)
            skips.append(x)
        if i < nb - 1:
            x = downsample(P, f"down_blocks.{i}.downsamplers.0.conv", x)
            skips.append(x)
    x = resnet(P, "mid_block.resnets.0.", x, emb, g, eps)
    x = transformer2d(P, "mid_block.attentions.0.", x, ctx, cfg.attention_head_dim[-1], cfg.transformer_layers_per_block[-1], g, cfg.use_linear_projection)
    x = resnet(P, "mid_block.resnets.1.", x, emb, g, eps)
    return skips, x, emb


def unet_forward(P: Dict[str, torch.Tensor], cfg: UNetConfig, sample, timesteps, encoder_hidden_states, added_cond_kwargs=None):
    """UNet2DConditionModel.forward -> [B, out_channels, H, W]"""
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    skips, x, emb = unet_down_mid(P, cfg, sample, timesteps, encoder_hidden_states, added_cond_kwargs)
    skips = list(skips)
    ctx = encoder_hidden_states.to(sample.dtype)
    nb = len(cfg.block_out_channels)
    for i, typ in enumerate(cfg.up_block_types):
        ri = nb - 1 - i                                   # the up path walks the channel list in reverse
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet(P, f"up_blocks.{i}.resnets.{j}.", x, emb, g, eps)
            if typ.startswith("CrossAttn"):
                x = transformer2d(P, f"up_blocks.{i}.attentions.{j}.", x, ctx, cfg.attention_head_dim[ri], cfg.transformer_layers_per_block[ri], g,
                                  cfg.use_linear_projection)
        if i < nb - 1:
            x = upsample(P, f"up_blocks.{i}.upsamplers.0.conv", x)
    x = F.silu(F.group_norm(x, g, P["conv_norm_out.weight"], P["conv_norm_out.bias"], eps))
    return _conv(x, P, "conv_out")


from tools.flop_count import unet_flops_fwd  # noqa: E402,F401  (one definition, shared with bench.py)


def init_params(cfg: UNetConfig, seed: int = 0, std_scale: float = 1.0, shapes_only: bool = False) -> Dict[str, torch.Tensor]:
    """Random weights under diffusers' names and shapes, walked in the order unet_forward consumes them (fan-in scaled normal, bias 0.02 N(0,1),
    norm weights 1 + 0.1 N(0,1)).  For CPU baselines / self-contained oracle runs; parity tests take their weights from the model under test."""
    g = torch.Generator().manual_seed(seed)
    P: Dict[str, torch.Tensor] = {}

    def rnd(*shape):                                     # shapes_only: meta tensors (names + shapes, no memory) for parameter accounting
        return torch.empty(*shape, device="meta") if shapes_only else torch.randn(*shape, generator=g)

    def lin(name, i, o, bias=True):
        P[name + ".weight"] = rnd(o, i) * (std_scale / math.sqrt(i))
        if bias:
            P[name + ".bias"] = 0.02 * rnd(o)

    def conv(name, i, o, k=3):
        P[name + ".weight"] = rnd(o, i, k, k) * (std_scale / math.sqrt(i * k * k))
        P[name + ".bias"] = 0.02 * rnd(o)

    def norm(name, c):
        P[name + ".weight"] = 1.0 + 0.1 * rnd(c)
        P[name + ".bias"] = 0.02 * rnd(c)

    ch = cfg.block_out_channels
    nb, temb = len(ch), 4 * ch[0]

    def res(p, ci, co):
        norm(p + "norm1", ci); conv(p + "conv1", ci, co); lin(p + "time_emb_proj", temb, co); norm(p + "norm2", co); conv(p + "conv2", co, co)
        if ci != co:
            conv(p + "conv_shortcut", ci, co, 1)

    def tr(p, c, n_layers):
        norm(p + "norm", c)
        (lin if cfg.use_linear_projection else (lambda n_, i, o: conv(n_, i, o, 1)))(p + "proj_in", c, c)
        for k in range(n_layers):
            b = f"{p}transformer_blocks.{k}."
            for a, kv in (("attn1.", c), ("attn2.", cfg.cross_attention_dim)):
                lin(b + a + "to_q", c, c, bias=False); lin(b + a + "to_k", kv, c, bias=False); lin(b + a + "to_v", kv, c, bias=False); lin(b + a + "to_out.0", c, c)
            for n_ in ("norm1", "norm2", "norm3"):
                norm(b + n_, c)
            lin(b + "ff.net.0.proj", c, 8 * c); lin(b + "ff.net.2", 4 * c, c)
        (lin if cfg.use_linear_projection else (lambda n_, i, o: conv(n_, i, o, 1)))(p + "proj_out", c, c)

    lin("time_embedding.linear_1", ch[0], temb); lin("time_embedding.linear_2", temb, temb)
    if cfg.addition_embed_type == "text_time":
        lin("add_embedding.linear_1", cfg.projection_class_embeddings_input_dim, temb); lin("add_embedding.linear_2", temb, temb)
    conv("conv_in", cfg.in_channels, ch[0])
    skip, cin = [ch[0]], ch[0]
    for i, typ in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            res(f"down_blocks.{i}.resnets.{j}.", cin, ch[i])
            cin = ch[i]
            if typ.startswith("CrossAttn"):
                tr(f"down_blocks.{i}.attentions.{j}.", cin, cfg.transformer_layers_per_block[i])
            skip.append(cin)
        if i < nb - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", cin, cin)
            skip.append(cin)
    res("mid_block.resnets.0.", cin, cin); tr("mid_block.attentions.0.", cin, cfg.transformer_layers_per_block[-1]); res("mid_block.resnets.1.", cin, cin)
    for i, typ in enumerate(cfg.up_block_types):
        ri = nb - 1 - i
        for j in range(cfg.layers_per_block + 1):
            res(f"up_blocks.{i}.resnets.{j}.", cin + skip.pop(), ch[ri])
            cin = ch[ri]
            if typ.startswith("CrossAttn"):
                tr(f"up_blocks.{i}.attentions.{j}.", cin, cfg.transformer_layers_per_block[ri])
        if i < nb - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", cin, cin)
    norm("conv_norm_out", cin)
    conv("conv_out", cin, cfg.out_channels)
    return P
