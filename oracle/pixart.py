"""Plain-torch CPU restatement of the PixArt-Sigma DiT forward and its ControlNet-Transformer wrapper (autograd = backward).  TEST INFRASTRUCTURE.

Control flow follows the reference's in-tree files:
    PixArtTransformer2DModel.forward                     simpletuner/helpers/models/pixart/transformer.py:499-788 (mask -> bias :566-568, blocks, final
                                                         modulation + proj_out + unpatchify "nhwpqc->nchpwq")
    ada_norm_single block arithmetic                     .../pixart/transformer.py:95-145 (table + t -> shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp;
                                                         cross-attention WITHOUT a pre-norm; gates on self-attention and MLP only)
    PixArtSigmaControlNetAdapterBlock / ...TransformerModel   .../pixart/controlnet.py:17-110, 166-326 (zero-init before_proj on block 0 input,
                                                         copied blocks, zero-init after_proj, residual into the trunk BEFORE trunk blocks 1..N)
    PixartSigma._model_predict_single / _controlnet_predict_single   .../pixart/model.py:274-319, 399-458 (chunk(2, dim=1)[0] drops the learned variance)
Leaf modules are diffusers' (un-vendored): PatchEmbed (conv p=2 + 2-D sincos table with interpolation_scale / base_size),
AdaLayerNormSingle (PixArtAlphaCombinedTimestepSizeEmbeddings + SiLU + Linear D->6D), PixArtAlphaTextProjection (Linear, GELU-tanh, Linear),
Attention (bias on q/k/v/out), FeedForward("gelu-approximate").
PINNED (round 3): this file reproduces, to <= 1e-5 in fp32 (outputs and every gradient), the outputs of the reference's OWN model files executed in the
build container over leaf-module shims (tools/ref_shim.py, tools/gen_ref_models.py -> tests/golden/ref_pixart_model.pt; tests/test_ref_models_cpu.py).
The control flow above the leaves is therefore pinned to executed reference code; the leaves (Linear / LayerNorm / SiLU compositions of diffusers, which is
absent from /root/reference) remain restatements, partly cross-checked against in-tree vendored copies.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .flux import timestep_proj


@dataclass
class PixArtConfig:
    # PixArt-Sigma XL/2 (pixart/transformer.py:210-230)
    num_attention_heads: int = 16
    attention_head_dim: int = 72
    in_channels: int = 4
    out_channels: int = 8
    num_layers: int = 28
    cross_attention_dim: int = 1152
    sample_size: int = 128
    patch_size: int = 2
    caption_channels: int = 4096
    use_additional_conditions: Optional[bool] = None
    interpolation_scale: Optional[float] = None

    @property
    def D(self):
        return self.num_attention_heads * self.attention_head_dim

    @property
    def additional(self):
        return (self.sample_size == 128) if self.use_additional_conditions is None else self.use_additional_conditions

    @property
    def interp(self):
        return self.interpolation_scale if self.interpolation_scale is not None else max(self.sample_size // 64, 1)


def sincos_2d_hw(embed_dim: int, h: int, w: int, base_size: int, interpolation_scale: float) -> torch.Tensor:
    """diffusers get_2d_sincos_pos_embed for a (h, w) grid: [h*w, embed_dim] (first half from the w coordinate, sin then cos per half)"""
    gh = (torch.arange(h, dtype=torch.float32) / (h / base_size) / interpolation_scale).double()
    gw = (torch.arange(w, dtype=torch.float32) / (w / base_size) / interpolation_scale).double()
    cw = gw[None, :].expand(h, w).reshape(-1)
    chh = gh[:, None].expand(h, w).reshape(-1)

    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
        out = pos[:, None] * omega[None, :]
        return torch.cat([out.sin(), out.cos()], dim=1)

    return torch.cat([one_d(embed_dim // 2, cw), one_d(embed_dim // 2, chh)], dim=1).float()


def _lin(x, P, name):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def patch_embed(P, cfg: PixArtConfig, x, prefix="pos_embed."):
    B, C, H, W = x.shape
    h, w = H // cfg.patch_size, W // cfg.patch_size
    t = F.conv2d(x, P[prefix + "proj.weight"], P[prefix + "proj.bias"], stride=cfg.patch_size).flatten(2).transpose(1, 2)
    pos = sincos_2d_hw(cfg.D, h, w, cfg.sample_size // cfg.patch_size, cfg.interp).to(t)      # table built in float64 on the host; the oracle itself may run on any device
    return t + pos[None]


def adaln_single(P, cfg: PixArtConfig, timestep, resolution, aspect_ratio, B, dtype):
    """-> (linear(silu(emb)) [B,6D], emb [B,D]); TOKENWISE timesteps [B, S] (pixart/transformer.py:790-850 `_embed_timesteps`): ([B,S,6D], [B,S,D]) — one
    timestep embedding per token, the size conditions shared by a sample's tokens"""
    def tee(x, p):
        return _lin(F.silu(_lin(x, P, p + ".linear_1")), P, p + ".linear_2")
    if timestep.ndim == 2:
        emb = tee(timestep_proj(timestep.reshape(-1).float(), 256).to(dtype), "adaln_single.emb.timestep_embedder").view(B, timestep.shape[1], -1)
    else:
        emb = tee(timestep_proj(timestep.expand(B), 256).to(dtype), "adaln_single.emb.timestep_embedder")
    if cfg.additional:
        r = tee(timestep_proj(resolution.flatten().float(), 256).to(dtype), "adaln_single.emb.resolution_embedder").reshape(B, -1)
        a = tee(timestep_proj(aspect_ratio.flatten().float(), 256).to(dtype), "adaln_single.emb.aspect_ratio_embedder").reshape(B, -1)
        size = torch.cat([r, a], dim=1)
        emb = emb + (size[:, None] if emb.ndim == 3 else size)
    return _lin(F.silu(emb), P, "adaln_single.linear"), emb


class _Fp8NativeLinearFn(torch.autograd.Function):
    """_Fp8NativeLinearFn (fp8_native.py:33-111): forward = e5m2 per-call activation x e4m3 row-scaled weight, fp32 accumulate, bf16 result; backward = grad_output @ the
    DEQUANTISED weight (:104-111) — the quantisers are not differentiated, the base weight gets no gradient"""

    @staticmethod
    def forward(ctx, x2, w, b):
        from . import train_math as TM
        q, sc = TM.fp8_quantize_weight(w.to(torch.bfloat16))
        xq, sa = TM.fp8_quantize_act(x2.to(torch.bfloat16))
        ctx.save_for_backward(q, sc)
        return TM.fp8_linear(xq, sa, q, sc, None if b is None else b.to(torch.bfloat16)).float()

    @staticmethod
    def backward(ctx, g):
        q, sc = ctx.saved_tensors
        w = q.to(torch.bfloat16) * sc.to(torch.bfloat16).unsqueeze(1)          # `weight.to(grad_output.dtype) * weight_scale.to(grad_output.dtype).unsqueeze(1)`, bf16 in training
        return g @ w.float(), None, None


def lin_fp8(x, P, name):
    """Fp8NativeLinear (fp8_native.py:25-119): e4m3 row-scaled weight, e5m2 per-call activation, fp32 accumulate, bf16 result; differentiable in x as the reference's
    autograd Function is (adapters over an fp8-native trunk)"""
    shp = x.shape
    return _Fp8NativeLinearFn.apply(x.reshape(-1, shp[-1]), P[name + ".weight"], P.get(name + ".bias")).reshape(*shp[:-1], -1)


def _attn(P, p, x, ctx, H, bias=None, _lin=_lin):
    B, S, D = x.shape
    q, k, v = _lin(x, P, p + "to_q"), _lin(ctx, P, p + "to_k"), _lin(ctx, P, p + "to_v")
    d = D // H
    q, k, v = (t.view(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
    def rows(qc):
        s = qc @ k.transpose(-1, -2) / math.sqrt(d)
        if bias is not None:
            s = s + bias[:, None, None, :]
        return s.softmax(-1) @ v
    # softmax is per query row: 4096-query slabs give the same values and keep the fp32 score matrix of a 16384-token (2K) self-attention at 4 GiB a slab
    o = rows(q) if q.shape[2] * k.shape[2] <= (1 << 26) else torch.cat([rows(qc) for qc in q.split(4096, dim=2)], dim=2)
    return _lin(o.transpose(1, 2).reshape(B, S, D), P, p + "to_out.0")


def block(P, p, cfg: PixArtConfig, h, ctx, ctx_bias, t6, _lin=_lin):
    """pixart/transformer.py:95-145 with timestep [B, 6D]; `_lin` = the Linear implementation of the block (plain, or lin_fp8)"""
    B, S, D = h.shape
    if t6.ndim == 3:        # tokenwise (:82-97): one modulation row per token
        mod = P[p + "scale_shift_table"][None, None] + t6.reshape(B, S, 6, D)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (t.squeeze(2) for t in mod.chunk(6, dim=2))
    else:
        mod = P[p + "scale_shift_table"][None] + t6.reshape(B, 6, D)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6, dim=1)
    n = F.layer_norm(h, (D,), eps=1e-6) * (1 + scale_msa) + shift_msa
    h = gate_msa * _attn(P, p + "attn1.", n, n, cfg.num_attention_heads, _lin=_lin) + h
    h = _attn(P, p + "attn2.", h, ctx, cfg.num_attention_heads, ctx_bias, _lin=_lin) + h
    n = F.layer_norm(h, (D,), eps=1e-6) * (1 + scale_mlp) + shift_mlp
    ff = _lin(F.gelu(_lin(n, P, p + "ff.net.0.proj"), approximate="tanh"), P, p + "ff.net.2")
    return gate_mlp * ff + h


def _head(P, cfg: PixArtConfig, h, emb, hh, ww):
    D = cfg.D
    if emb.ndim == 3:       # tokenwise (:749-753)
        shift, scale = ((P["scale_shift_table"][None, None] + emb[:, :, None]).chunk(2, dim=2))
        shift, scale = shift.squeeze(2), scale.squeeze(2)
    else:
        shift, scale = (P["scale_shift_table"][None] + emb[:, None]).chunk(2, dim=1)
    h = F.layer_norm(h, (D,), eps=1e-6) * (1 + scale) + shift
    h = _lin(h, P, "proj_out")
    p = cfg.patch_size
    h = h.reshape(-1, hh, ww, p, p, cfg.out_channels)
    return torch.einsum("nhwpqc->nchpwq", h).reshape(-1, cfg.out_channels, hh * p, ww * p)


def _prep(P, cfg, latents, enc, mask, timestep, resolution, aspect_ratio):
    B = latents.shape[0]
    dt = latents.dtype
    bias = None if mask is None else (1 - mask.to(dt)) * -10000.0                         # pixart/transformer.py:566-568
    h = patch_embed(P, cfg, latents)
    t6, emb = adaln_single(P, cfg, timestep, resolution, aspect_ratio, B, dt)
    ctx = _lin(F.gelu(_lin(enc.to(dt), P, "caption_projection.linear_1"), approximate="tanh"), P, "caption_projection.linear_2")
    return h, t6, emb, ctx, bias


def lora_targets(cfg: PixArtConfig):
    """pixart/model.py:59 DEFAULT_LORA_TARGET = to_k, to_q, to_v, to_out.0 — peft matches by suffix: attn1 and attn2 of every block"""
    return [f"transformer_blocks.{i}.{a}.{n}" for i in range(cfg.num_layers) for a in ("attn1", "attn2") for n in ("to_q", "to_k", "to_v", "to_out.0")]


def pixart_forward(P: Dict[str, torch.Tensor], cfg: PixArtConfig, latents, enc, mask, timestep, resolution=None, aspect_ratio=None, fp8_blocks: bool = False,
                   lora=None, lora_scale: float = 1.0, tread=None):
    """PixArtTransformer2DModel.forward -> [B, out_channels, H, W]; fp8_blocks: the transformer blocks' Linears in the fp8-native form.
    lora = {module name: (A [r, in], B [out, r])}: peft adapters y += scale * B A x on the named Linears.  tread: as oracle.flux.flux_forward — token routing between
    two block indices (pixart/transformer.py:487-489 `set_router` and the routed span of the block loop) with the router's permutations replayed."""
    from .flux import tread_end, tread_start
    hh, ww = latents.shape[-2] // cfg.patch_size, latents.shape[-1] // cfg.patch_size
    h, t6, emb, ctx, bias = _prep(P, cfg, latents, enc, mask, timestep, resolution, aspect_ratio)
    base = lin_fp8 if fp8_blocks else _lin

    def lin(x, P_, name):
        y = base(x, P_, name)
        if lora is not None and name in lora:
            A, B_ = lora[name]
            y = y + lora_scale * ((x @ A.t().to(x.dtype)) @ B_.t().to(x.dtype))
        return y

    routes = [dict(r, start_layer_idx=r["start_layer_idx"] % cfg.num_layers, end_layer_idx=r["end_layer_idx"] % cfg.num_layers) for r in (tread or {}).get("routes", [])]
    infos = (tread or {}).get("mask_infos", [])
    ptr, info, saved = 0, None, None
    for i in range(cfg.num_layers):
        if ptr < len(routes) and i == routes[ptr]["start_layer_idx"]:
            info, saved = infos[ptr], h
            h = tread_start(h, info)
        h = block(P, f"transformer_blocks.{i}.", cfg, h, ctx, bias, t6, _lin=lin)
        if info is not None and i == routes[ptr]["end_layer_idx"]:
            h = tread_end(h, info, saved)
            info, saved, ptr = None, None, ptr + 1
    return _head(P, cfg, h, emb, hh, ww)


def controlnet_forward(P, C, cfg: PixArtConfig, n_ctrl: int, latents, cond, enc, mask, timestep, resolution=None, aspect_ratio=None, checkpoint: bool = False):
    """PixArtSigmaControlNetTransformerModel.forward (pixart/controlnet.py:208-326).  P: trunk weights; C: adapter weights
    (`controlnet_blocks.{i}.before_proj|transformer_block.*|after_proj`).  checkpoint=True: every block re-run in the backward (torch.utils.checkpoint,
    the reference's gradient-checkpointing branch :262-296) — same values, one block's activations alive at a time (2K latents as an fp32 checker)."""
    if checkpoint:
        import torch.utils.checkpoint as _ck
        run = lambda *a: _ck.checkpoint(block, *a, use_reentrant=False)
    else:
        run = block
    hh, ww = latents.shape[-2] // cfg.patch_size, latents.shape[-1] // cfg.patch_size
    h, t6, emb, ctx, bias = _prep(P, cfg, latents, enc, mask, timestep, resolution, aspect_ratio)
    cs = patch_embed(P, cfg, cond)
    for i in range(cfg.num_layers):
        if 0 < i <= n_ctrl:
            p = f"controlnet_blocks.{i - 1}."
            if i == 1:
                cs = h + _lin(cs, C, p + "before_proj")
            cs = run(C, p + "transformer_block.", cfg, cs, ctx, bias, t6)
            h = h + _lin(cs, C, p + "after_proj")
        h = run(P, f"transformer_blocks.{i}.", cfg, h, ctx, bias, t6)
    return _head(P, cfg, h, emb, hh, ww)


def param_shapes(cfg: PixArtConfig) -> Dict[str, tuple]:
    """diffusers' PixArtTransformer2DModel state-dict names -> shapes, as the functions above consume them.  With use_additional_conditions=False
    the total is 610,856,096 — the published size of PixArt-Sigma-XL-2 (tests/test_oracles_cpu.py)."""
    D, sh = cfg.D, {}

    def lin(name, i, o):
        sh[name + ".weight"], sh[name + ".bias"] = (o, i), (o,)

    sh["pos_embed.proj.weight"], sh["pos_embed.proj.bias"] = (D, cfg.in_channels, cfg.patch_size, cfg.patch_size), (D,)
    lin("adaln_single.emb.timestep_embedder.linear_1", 256, D); lin("adaln_single.emb.timestep_embedder.linear_2", D, D)
    if cfg.additional:
        for n in ("resolution_embedder", "aspect_ratio_embedder"):
            lin(f"adaln_single.emb.{n}.linear_1", 256, D // 3); lin(f"adaln_single.emb.{n}.linear_2", D // 3, D // 3)
    lin("adaln_single.linear", D, 6 * D)
    lin("caption_projection.linear_1", cfg.caption_channels, D); lin("caption_projection.linear_2", D, D)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        sh[b + "scale_shift_table"] = (6, D)
        for a in ("attn1.", "attn2."):
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(b + a + n, cfg.cross_attention_dim if (a == "attn2." and n in ("to_k", "to_v")) else D, D)
        lin(b + "ff.net.0.proj", D, 4 * D); lin(b + "ff.net.2", 4 * D, D)
    sh["scale_shift_table"] = (2, D)
    lin("proj_out", D, cfg.patch_size * cfg.patch_size * cfg.out_channels)
    return sh


from tools.flop_count import pixart_flops_fwd  # noqa: E402,F401
