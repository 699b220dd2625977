"""Plain-torch CPU restatement of the SD3 MMDiT forward (autograd supplies the backward).  TEST INFRASTRUCTURE.

Control flow follows the reference's in-tree transformer:
    SD3Transformer2DModel.forward             simpletuner/helpers/models/sd3/transformer.py:560-911 (unpatchify :879-902)
    _sd3_apply_joint_transformer_block        .../sd3/transformer.py:145-241   (context_pre_only last block :174-176, :214-215)
    _sd3_apply_ada_layer_norm_zero / _continuous  .../sd3/transformer.py:126-142 (chunk orders: shift,scale,gate,... / scale,shift)
    SD3._model_predict_single                 simpletuner/helpers/models/sd3/model.py:540-570 (timestep passed in 0..1000)
Leaf modules come from diffusers (>=0.36, un-vendored): PatchEmbed (Conv2d k=p, s=p + centre-cropped 2-D
sincos table of pos_embed_max_size^2 positions), CombinedTimestepTextProjEmbeddings, JointAttnProcessor2_0 (joint sequence =
[sample || context], no RoPE, optional RMSNorm on q/k), FeedForward("gelu-approximate").  The sincos table follows the public
diffusers `get_2d_sincos_pos_embed` (grid / (grid_size/base_size) / interpolation_scale, w-axis first): UNCORROBORATED in-tree —
with a real checkpoint the table is a loaded buffer (`pos_embed.pos_embed`), so only the CROP affects parity.
PINNED (round 3): this file reproduces, to <= 1e-5 in fp32 (outputs and every gradient), the outputs of the reference's OWN model files executed in the
build container over leaf-module shims (tools/ref_shim.py, tools/gen_ref_models.py -> tests/golden/ref_sd3_model.pt; tests/test_ref_models_cpu.py).
The control flow above the leaves is therefore pinned to executed reference code; the leaves (Linear / LayerNorm / SiLU compositions of diffusers, which is
absent from /root/reference) remain restatements, partly cross-checked against in-tree vendored copies.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .flux import _heads, layer_norm, linear, mlp_embed, rms_norm, sdpa, timestep_proj


@dataclass
class SD3Config:
    # SD3-Medium: 24 layers, 24 x 64 heads (D = 1536), patch 2, 16 latent channels, pooled 2048, ctx 4096
    sample_size: int = 128
    patch_size: int = 2
    in_channels: int = 16
    num_layers: int = 24
    attention_head_dim: int = 64
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 2048
    out_channels: int = 16
    pos_embed_max_size: int = 192
    qk_norm: Optional[str] = None           # "rms_norm" for SD3.5
    dual_attention_layers: Tuple[int, ...] = ()   # SD3.5-medium: (0..12); sd3/transformer.py:295, 155-165, 190-197

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


def sincos_2d(embed_dim: int, grid_size: int, base_size: int, interpolation_scale: float = 1.0) -> torch.Tensor:
    """diffusers get_2d_sincos_pos_embed (numpy form) -> [grid_size^2, embed_dim] fp32"""
    gh = np.arange(grid_size, dtype=np.float32) / (grid_size / base_size) / interpolation_scale
    gw = np.arange(grid_size, dtype=np.float32) / (grid_size / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)   # w goes first

    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float()


def cropped_pos_embed(table: torch.Tensor, max_size: int, h: int, w: int) -> torch.Tensor:
    """PatchEmbed.cropped_pos_embed: centre crop of the [max,max] grid to [h,w] token rows -> [h*w, D]"""
    top, left = (max_size - h) // 2, (max_size - w) // 2
    t = table.view(max_size, max_size, -1)[top:top + h, left:left + w]
    return t.reshape(h * w, -1)


def param_shapes(cfg: SD3Config) -> Dict[str, Tuple[int, ...]]:
    D, d, p = cfg.inner_dim, cfg.attention_head_dim, cfg.patch_size
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, out_f, in_f):
        s[name + ".weight"] = (out_f, in_f)
        s[name + ".bias"] = (out_f,)

    s["pos_embed.proj.weight"] = (D, cfg.in_channels, p, p)
    s["pos_embed.proj.bias"] = (D,)
    lin("time_text_embed.timestep_embedder.linear_1", D, 256)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, cfg.pooled_projection_dim)
    lin("time_text_embed.text_embedder.linear_2", D, D)
    lin("context_embedder", D, cfg.joint_attention_dim)
    for i in range(cfg.num_layers):
        last = i == cfg.num_layers - 1
        pfx = f"transformer_blocks.{i}."
        dual = i in cfg.dual_attention_layers
        lin(pfx + "norm1.linear", (9 if dual else 6) * D, D)
        lin(pfx + "norm1_context.linear", (2 if last else 6) * D, D)
        if dual:
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(pfx + "attn2." + n, D, D)
            if cfg.qk_norm == "rms_norm":
                for n in ("norm_q", "norm_k"):
                    s[pfx + f"attn2.{n}.weight"] = (d,)
        names = ["to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0"] + ([] if last else ["to_add_out"])
        for n in names:
            lin(pfx + "attn." + n, D, D)
        if cfg.qk_norm == "rms_norm":
            for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                s[pfx + f"attn.{n}.weight"] = (d,)
        lin(pfx + "ff.net.0.proj", 4 * D, D)
        lin(pfx + "ff.net.2", D, 4 * D)
        if not last:
            lin(pfx + "ff_context.net.0.proj", 4 * D, D)
            lin(pfx + "ff_context.net.2", D, 4 * D)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", p * p * cfg.out_channels, D)
    return s


def init_params(cfg: SD3Config, seed: int = 42, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if "norm_q" in name or "norm_k" in name or "norm_added" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        out[name] = t.to(dtype)
    out["pos_embed.pos_embed"] = sincos_2d(cfg.inner_dim, cfg.pos_embed_max_size, cfg.sample_size // cfg.patch_size).to(dtype)[None]
    return out


def lora_targets(cfg: SD3Config):
    """sd3/model.py:122 DEFAULT_LORA_TARGET = to_k, to_q, to_v, to_out.0 (the sample-stream attention projections)"""
    t = []
    for i in range(cfg.num_layers):
        t += [f"transformer_blocks.{i}.attn." + n for n in ("to_q", "to_k", "to_v", "to_out.0")]
    return t


def joint_block(P, cfg: SD3Config, i: int, hidden, enc, temb, lora=None, lora_scale=1.0, temb_context=None):
    """sd3/transformer.py:145-241.  Returns (enc | None, hidden).  temb [B, D]; or TOKENWISE, temb [B, S_img, D] with temb_context [B, D] = its mean over the
    tokens (:126-142 `_sd3_apply_ada_layer_norm_zero`, :680-685): the image stream's shift / scale / gate rows are then per token, the context stream's per sample."""
    pfx = f"transformer_blocks.{i}."
    last = i == cfg.num_layers - 1
    H = cfg.num_attention_heads
    tokenwise = temb.ndim == 3
    st_i = F.silu(temb)
    st = F.silu(temb_context) if tokenwise else st_i
    dual = i in cfg.dual_attention_layers
    if dual:            # SD35AdaLayerNormZeroX: 9 chunks, the last three modulate the input of the second (image-only) attention
        if tokenwise:
            raise ValueError("tokenwise timesteps with SD3.5 dual-attention blocks: the reference hands the [B, S, D] embedding to SD35AdaLayerNormZeroX, "
                             "which chunks dim 1 (the tokens) — not a defined computation")
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp, shift_msa2, scale_msa2, gate_msa2 = linear(st_i, P, pfx + "norm1.linear").chunk(9, dim=1)
        n2a = layer_norm(hidden) * (1 + scale_msa2[:, None]) + shift_msa2[:, None]
    else:
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = linear(st_i, P, pfx + "norm1.linear").chunk(6, dim=-1)
    if not tokenwise:
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (t[:, None] for t in (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp))
    n = layer_norm(hidden) * (1 + scale_msa) + shift_msa
    if last:
        c_scale, c_shift = linear(st, P, pfx + "norm1_context.linear").chunk(2, dim=1)      # AdaLayerNormContinuous: scale first
        cn = layer_norm(enc) * (1 + c_scale[:, None]) + c_shift[:, None]
    else:
        c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = linear(st, P, pfx + "norm1_context.linear").chunk(6, dim=1)
        cn = layer_norm(enc) * (1 + c_scale_msa[:, None]) + c_shift_msa[:, None]
    a = pfx + "attn."
    q, k, v = (_heads(linear(n, P, a + nm, lora, lora_scale), H) for nm in ("to_q", "to_k", "to_v"))
    cq, ck, cv = (_heads(linear(cn, P, a + nm, lora, lora_scale), H) for nm in ("add_q_proj", "add_k_proj", "add_v_proj"))
    if cfg.qk_norm == "rms_norm":
        q = rms_norm(q, P[a + "norm_q.weight"]); k = rms_norm(k, P[a + "norm_k.weight"])
        cq = rms_norm(cq, P[a + "norm_added_q.weight"]); ck = rms_norm(ck, P[a + "norm_added_k.weight"])
    q = torch.cat([q, cq], dim=2); k = torch.cat([k, ck], dim=2); v = torch.cat([v, cv], dim=2)          # [sample || context]
    o = sdpa(q, k, v)
    B, _, S, _ = o.shape
    o = o.transpose(1, 2).reshape(B, S, -1)
    Si = hidden.shape[1]
    io, co = o[:, :Si], o[:, Si:]
    hidden = hidden + gate_msa * linear(io, P, a + "to_out.0", lora, lora_scale)
    if dual:            # sd3/transformer.py:190-197: attn2 = self-attention over the image tokens only, added after the joint-attention residual
        a2 = pfx + "attn2."
        q2, k2, v2 = (_heads(linear(n2a, P, a2 + nm, lora, lora_scale), H) for nm in ("to_q", "to_k", "to_v"))
        if cfg.qk_norm == "rms_norm":
            q2 = rms_norm(q2, P[a2 + "norm_q.weight"]); k2 = rms_norm(k2, P[a2 + "norm_k.weight"])
        o2 = sdpa(q2, k2, v2).transpose(1, 2).reshape(B, Si, -1)
        hidden = hidden + gate_msa2[:, None] * linear(o2, P, a2 + "to_out.0", lora, lora_scale)
    n2 = layer_norm(hidden) * (1 + scale_mlp) + shift_mlp
    hidden = hidden + gate_mlp * linear(F.gelu(linear(n2, P, pfx + "ff.net.0.proj"), approximate="tanh"), P, pfx + "ff.net.2")
    if last:
        return None, hidden
    enc = enc + c_gate_msa[:, None] * linear(co, P, a + "to_add_out", lora, lora_scale)
    cn2 = layer_norm(enc) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
    enc = enc + c_gate_mlp[:, None] * linear(F.gelu(linear(cn2, P, pfx + "ff_context.net.0.proj"), approximate="tanh"), P, pfx + "ff_context.net.2")
    return enc, hidden


def sd3_forward(P, cfg: SD3Config, latents, encoder_hidden_states, pooled_projections, timestep, lora=None, lora_scale: float = 1.0, tread=None,
                checkpoint: bool = False):
    """sd3/transformer.py:560-911.  latents [B,16,H,W]; timestep [B] in 0..1000.  Returns [B,16,H,W].
    checkpoint: re-run every joint block in the backward instead of keeping its activations (torch.utils.checkpoint, non-reentrant): the same fp32 arithmetic, so
    that a multi-step full-depth trajectory fits next to the model under test (as oracle.flux.flux_forward(checkpoint=True)).
    tread: as oracle.flux.flux_forward — routing over the image tokens between two block indices (:694-706, 796-803; no RoPE to re-route)."""
    from .flux import tread_end, tread_start
    B, C, Hh, Ww = latents.shape
    p = cfg.patch_size
    h, w = Hh // p, Ww // p
    D = cfg.inner_dim
    x = F.conv2d(latents, P["pos_embed.proj.weight"], P["pos_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)   # [B, hw, D]
    hidden = x + cropped_pos_embed(P["pos_embed.pos_embed"][0], cfg.pos_embed_max_size, h, w)[None].to(x.dtype)
    dt = pooled_projections.dtype
    temb_context = None
    if timestep.ndim == 2:          # TOKENWISE timesteps [B, S_img] (:61-75 `_sd3_tokenwise_conditioning`, :625-626, :680-685): one conditioning row per image token
        if timestep.shape != (B, h * w):
            raise ValueError(f"SD3 tokenwise timesteps expected sequence length {h * w}, got {timestep.shape[1]}.")
        temb = mlp_embed(timestep_proj(timestep.reshape(-1).float()).to(dt), P, "time_text_embed.timestep_embedder").view(B, h * w, -1) + \
            mlp_embed(pooled_projections, P, "time_text_embed.text_embedder")[:, None]
        temb_context = temb.mean(dim=1)
        if tread:
            raise ValueError("tokenwise timesteps under TREAD routing are not restated")
    else:
        temb = mlp_embed(timestep_proj(timestep.float()).to(dt), P, "time_text_embed.timestep_embedder") + \
            mlp_embed(pooled_projections, P, "time_text_embed.text_embedder")
    enc = linear(encoder_hidden_states, P, "context_embedder")
    routes = [dict(r, start_layer_idx=r["start_layer_idx"] % cfg.num_layers, end_layer_idx=r["end_layer_idx"] % cfg.num_layers)
              for r in (tread or {}).get("routes", [])]
    infos = (tread or {}).get("mask_infos", [])
    ptr, info, saved = 0, None, None
    for i in range(cfg.num_layers):
        if ptr < len(routes) and i == routes[ptr]["start_layer_idx"]:
            info, saved = infos[ptr], hidden
            hidden = tread_start(hidden, info)
        if checkpoint and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint as _ckpt
            enc, hidden = _ckpt(lambda h_, e_, t_, i_=i: joint_block(P, cfg, i_, h_, e_, t_, lora, lora_scale, temb_context), hidden, enc, temb, use_reentrant=False)
        else:
            enc, hidden = joint_block(P, cfg, i, hidden, enc, temb, lora, lora_scale, temb_context)
        if info is not None and i == routes[ptr]["end_layer_idx"]:
            hidden = tread_end(hidden, info, saved)
            info, saved, ptr = None, None, ptr + 1
    scale, shift = linear(F.silu(temb), P, "norm_out.linear").chunk(2, dim=-1)          # :876 norm_out takes temb_hidden: per token when tokenwise
    if temb.ndim == 2:
        scale, shift = scale[:, None], shift[:, None]
    hidden = layer_norm(hidden) * (1 + scale) + shift
    out = linear(hidden, P, "proj_out")                                    # [B, hw, p*p*C]
    out = out.reshape(B, h, w, p, p, cfg.out_channels)
    return torch.einsum("nhwpqc->nchpwq", out).reshape(B, cfg.out_channels, h * p, w * p)


def train_flops_per_image(cfg: SD3Config, S_img: int, S_txt: int, lora: bool = True) -> float:
    """per block 2*S*12D^2 (linears) + 4*S^2*D (attention) forward; LoRA step = 2x linears + 3x attention,
    full fine-tune = 3x both."""
    D, S = cfg.inner_dim, S_img + S_txt
    lin = cfg.num_layers * 2.0 * S * 12 * D * D
    att = cfg.num_layers * 4.0 * S * S * D
    return (2 * lin + 3 * att) if lora else 3 * (lin + att)
