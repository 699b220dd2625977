"""Plain-torch CPU restatement of the Flux.1 MMDiT forward (autograd supplies the backward).  TEST INFRASTRUCTURE.

Control flow follows the reference's in-tree transformer:
    FluxTransformer2DModel.forward        simpletuner/helpers/models/flux/transformer.py:940-1513
    FluxTransformerBlock.forward          .../flux/transformer.py:607-687   (_ffn_forward :563-605)
    FluxSingleTransformerBlock.forward    .../flux/transformer.py:473-510   (_ffn_forward :453-471)
    FluxAttnProcessor2_0.__call__         .../flux/transformer.py:116-224
    _apply_rotary_emb_anyshape            .../flux/transformer.py:73-98
    Flux._model_predict_single            simpletuner/helpers/models/flux/model.py:707-864
    pack/unpack/prepare_latent_image_ids  simpletuner/helpers/models/flux/__init__.py:25-63
Leaf modules come from diffusers (>=0.36, un-vendored; SURVEY.md Appendix A lists the in-tree corroboration):
    Timesteps / TimestepEmbedding / CombinedTimestep(Guidance)TextProjEmbeddings, AdaLayerNormZero(-Single/-Continuous),
    RMSNorm, FluxPosEmbed (theta 1e4, axes (16,56,56)), FeedForward("gelu-approximate"), Attention projections.
LoRA follows peft's LoraLayer: y = base(x) + scaling * lora_B(lora_A(x)), scaling = alpha / r
    (corroborated in-tree at simpletuner/helpers/training/quantisation/peft_workarounds.py:70-116).

Parameters are a flat dict keyed by the diffusers state-dict names, so the same dict initialises the HIP model.
PINNED (round 3): this file reproduces, to <= 1e-5 in fp32 (outputs and every gradient), the outputs of the reference's OWN model files executed in the
build container over leaf-module shims (tools/ref_shim.py, tools/gen_ref_models.py -> tests/golden/ref_flux_model.pt; tests/test_ref_models_cpu.py).
The control flow above the leaves is therefore pinned to executed reference code; the leaves (Linear / LayerNorm / SiLU compositions of diffusers, which is
absent from /root/reference) remain restatements, partly cross-checked against in-tree vendored copies.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class FluxConfig:
    # flux/transformer.py:725-742 defaults (Flux.1-dev sets guidance_embeds=True)
    in_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: Tuple[int, ...] = (16, 56, 56)

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


# ------------------------------------------------------------------------------------------------
# parameter construction (seed-deterministic synthetic weights)
# ------------------------------------------------------------------------------------------------
def param_shapes(cfg: FluxConfig) -> Dict[str, Tuple[int, ...]]:
    D, d = cfg.inner_dim, cfg.attention_head_dim
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, out_f, in_f):
        s[name + ".weight"] = (out_f, in_f)
        s[name + ".bias"] = (out_f,)

    lin("x_embedder", D, cfg.in_channels)
    lin("context_embedder", D, cfg.joint_attention_dim)
    lin("time_text_embed.timestep_embedder.linear_1", D, 256)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    if cfg.guidance_embeds:
        lin("time_text_embed.guidance_embedder.linear_1", D, 256)
        lin("time_text_embed.guidance_embedder.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, cfg.pooled_projection_dim)
    lin("time_text_embed.text_embedder.linear_2", D, D)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}."
        lin(p + "norm1.linear", 6 * D, D)
        lin(p + "norm1_context.linear", 6 * D, D)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(p + "attn." + n, D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[p + f"attn.{n}.weight"] = (d,)
        lin(p + "ff.net.0.proj", 4 * D, D)
        lin(p + "ff.net.2", D, 4 * D)
        lin(p + "ff_context.net.0.proj", 4 * D, D)
        lin(p + "ff_context.net.2", D, 4 * D)
    for i in range(cfg.num_single_layers):
        p = f"single_transformer_blocks.{i}."
        lin(p + "norm.linear", 3 * D, D)
        lin(p + "proj_mlp", 4 * D, D)
        lin(p + "proj_out", D, 5 * D)
        for n in ("to_q", "to_k", "to_v"):
            lin(p + "attn." + n, D, D)
        for n in ("norm_q", "norm_k"):
            s[p + f"attn.{n}.weight"] = (d,)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", cfg.in_channels, D)
    return s


def init_params(cfg: FluxConfig, seed: int = 42, dtype=torch.float32, std: float = 0.02, device="cpu") -> Dict[str, torch.Tensor]:
    """weights ~ N(0, std^2) scaled by 1/sqrt(fan_in/256) so activations stay O(1) through many blocks; biases small
    nonzero; RMSNorm weights ~ 1; AdaLN linears nonzero so gates pass gradient (true zero-init would block it)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm_added_q.weight") or name.endswith("norm_added_k.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[1]
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        out[name] = t.to(dtype=dtype, device=device)
    return out


def lora_targets(cfg: FluxConfig, which: str = "default"):
    """peft target modules for Flux (flux/model.py:65 DEFAULT_LORA_TARGET = to_k,to_q,to_v,to_out.0; flux/model.py:1249-1375 the `flux_lora_target` sets).
    'default' here = the attention projections of the image stream + single blocks; 'all' adds the context-stream projections, 'context' = only those;
    '+ffs' adds the feed-forward Linears (all+ffs: both streams' ff.net.* and the single blocks' proj_mlp / proj_out; context+ffs: ff_context.net.*);
    'tiny' / 'nano' = single_transformer_blocks.{7, 20}.proj_out / .7.proj_out (flux/model.py:1363-1375)."""
    if which in ("tiny", "nano"):
        blocks = (7, 20) if which == "tiny" else (7,)
        return [f"single_transformer_blocks.{i}.proj_out" for i in blocks if i < cfg.num_single_layers]
    emb = which.endswith("+embedder")          # all+ffs+embedder (flux/model.py:1320-1339): all+ffs + x_embedder
    which = which[:-len("+embedder")] if emb else which
    if which == "ai-toolkit":                  # flux/model.py:1340-1362: all+ffs + the AdaLN modulation Linears of every block
        mods = [f"transformer_blocks.{i}.{n}" for i in range(cfg.num_layers) for n in ("norm1.linear", "norm1_context.linear")] + \
               [f"single_transformer_blocks.{i}.norm.linear" for i in range(cfg.num_single_layers)]
        return mods + lora_targets(cfg, "all+ffs")
    base, ffs = (which[:-4], True) if which.endswith("+ffs") else (which, False)
    t = ["x_embedder"] if emb else []
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}."
        p = b + "attn."
        if base != "context":
            t += [p + n for n in ("to_q", "to_k", "to_v", "to_out.0")]
        if base in ("all", "context"):
            t += [p + n for n in ("add_q_proj", "add_k_proj", "add_v_proj", "to_add_out")]
        if ffs and base == "all":
            t += [b + n for n in ("ff.net.0.proj", "ff.net.2")]
        if ffs:
            t += [b + n for n in ("ff_context.net.0.proj", "ff_context.net.2")]
    if base != "context":
        for i in range(cfg.num_single_layers):
            b = f"single_transformer_blocks.{i}."
            t += [b + "attn." + n for n in ("to_q", "to_k", "to_v")]
            if ffs:
                t += [b + "proj_mlp", b + "proj_out"]
        if ffs:
            t += ["proj_out"]          # peft's suffix rule: the entry "proj_out" also names the model's own output projection
    return t


def init_lora(cfg: FluxConfig, params, rank: int, seed: int = 7, b_std: float = 1e-3, dtype=torch.float32, which="default"):
    """A ~ kaiming-uniform(a=sqrt(5)) (peft default), B ~ N(0, b_std^2) (true init is 0; nonzero so the delta is visible)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lora = {}
    for name in lora_targets(cfg, which):
        out_f, in_f = params[name + ".weight"].shape
        bound = 1.0 / math.sqrt(in_f)
        A = (torch.rand(rank, in_f, generator=g) * 2 - 1) * bound
        Bm = torch.randn(out_f, rank, generator=g) * b_std
        lora[name] = (A.to(dtype), Bm.to(dtype))
    return lora


# ------------------------------------------------------------------------------------------------
# leaves
# ------------------------------------------------------------------------------------------------
def timestep_proj(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) (fp32)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([emb.cos(), emb.sin()], dim=-1)  # flip_sin_to_cos=True -> cos first


def linear(x, P, name, lora=None, lora_scale: float = 1.0):
    y = F.linear(x, P[name + ".weight"], P.get(name + ".bias"))
    if lora is not None and name in lora:
        A, Bm = lora[name]
        y = y + lora_scale * F.linear(F.linear(x, A.to(x.dtype)), Bm.to(x.dtype))
    return y


def mlp_embed(x, P, prefix):  # TimestepEmbedding / PixArtAlphaTextProjection(act="silu")
    return linear(F.silu(linear(x, P, prefix + ".linear_1")), P, prefix + ".linear_2")


def time_text_embed(P, cfg, timestep, guidance, pooled):
    dt = pooled.dtype
    emb = mlp_embed(timestep_proj(timestep).to(dt), P, "time_text_embed.timestep_embedder")
    if cfg.guidance_embeds:
        emb = emb + mlp_embed(timestep_proj(guidance).to(dt), P, "time_text_embed.guidance_embedder")
    return emb + mlp_embed(pooled, P, "time_text_embed.text_embedder")


def layer_norm(x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), eps=eps)


def rms_norm(x, w, eps=1e-6):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def rope_tables(ids: torch.Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0):
    """FluxPosEmbed: per axis freqs = 1/theta^(arange(0,d,2)/d) in float64; cos/sin repeat_interleave(2) -> fp32 [S, sum(d)]."""
    cos_out, sin_out = [], []
    pos = ids.float()
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64)[: d // 2] / d))
        f = torch.outer(pos[:, i].to(torch.float64), freqs)
        cos_out.append(f.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(f.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


def apply_rope(x, cos, sin):
    """flux/transformer.py:73-98 (use_real, unbind_dim=-1): out = x*cos + stack[-x_imag, x_real]*sin, computed in >= fp32.
    cos / sin are [S, d] or, inside a TREAD route, per-sample [B, S, d] (:80-85)."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    ct = torch.promote_types(x.dtype, torch.float32)
    cos, sin = (t[:, None] if t.ndim == 3 else t[None, None] for t in (cos, sin))
    return (x.to(ct) * cos.to(ct) + rot.to(ct) * sin.to(ct)).to(x.dtype)


# ------------------------------------------------------------------------------------------------
# TREAD token routing (training/tread.py:58-159): the router's permutations are INPUTS here (recorded from / replayed into the
# reference), the gather / truncate / scatter-back arithmetic is restated
# ------------------------------------------------------------------------------------------------
def tread_start(x, info):
    """TREADRouter.start_route (tread.py:118-125): kept tokens first, truncated to K"""
    K = info["ids_keep"].shape[1]
    return torch.take_along_dim(x, info["ids_shuffle"].unsqueeze(-1).expand_as(x), dim=1)[:, :K]


def tread_end(routed, info, original):
    """TREADRouter.end_route (tread.py:127-159) with original_x: skipped tokens keep their pre-route values"""
    K = routed.shape[1]
    orig_shuf = torch.take_along_dim(original, info["ids_shuffle"].unsqueeze(-1).expand_as(original), dim=1)
    x_shuf = torch.cat([routed, orig_shuf[:, K:]], dim=1)
    return torch.take_along_dim(x_shuf, info["ids_restore"].unsqueeze(-1).expand_as(x_shuf), dim=1)


def tread_rope(cos, sin, T, info, K, B):
    """flux/transformer.py:1227-1241: rope rows = [text rows, batch-expanded | image rows shuffled like the tokens, first K]"""
    out = []
    for r in (cos, sin):
        txt = r[:T].unsqueeze(0).expand(B, -1, -1)
        img = r[T:].unsqueeze(0).expand(B, -1, -1)
        img = torch.take_along_dim(img, info["ids_shuffle"].unsqueeze(-1).expand_as(img), dim=1)[:, :K]
        out.append(torch.cat([txt, img], dim=1))
    return out


def sdpa(q, k, v, key_bias=None):
    scale = 1.0 / math.sqrt(q.shape[-1])
    s = (q @ k.transpose(-1, -2)) * scale
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    return s.softmax(-1) @ v


def prepare_latent_image_ids(height: int, width: int) -> torch.Tensor:
    """flux/__init__.py:48-63 -> [ (H/2)(W/2), 3 ] fp32"""
    ids = torch.zeros(height // 2, width // 2, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(height // 2)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(width // 2)[None, :]
    return ids.reshape(-1, 3).float()


def pack_latents(latents):
    B, C, H, W = latents.shape
    return latents.view(B, C, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // 2) * (W // 2), C * 4)


def unpack_latents(packed, H, W):
    """flux/__init__.py:34-45 with height/width already in latent units (vae_scale_factor folded)."""
    B, N, ch = packed.shape
    x = packed.view(B, H // 2, W // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(B, ch // 4, H, W)


# ------------------------------------------------------------------------------------------------
# blocks
# ------------------------------------------------------------------------------------------------
def _heads(x, H):
    B, S, D = x.shape
    return x.view(B, S, H, D // H).transpose(1, 2)


def _rows(t):
    """a modulation chunk as it multiplies [B, S, D] activations: [B, D] rows per sample broadcast over the tokens, [B, S, D] rows per token as they are"""
    return t if t.ndim == 3 else t[:, None]


def double_block(P, cfg, i, hidden, enc, temb, cos, sin, lora=None, lora_scale=1.0, key_bias=None, taps=None, temb_txt=None):
    """flux/transformer.py:607-687.  temb [B, D]; or TOKENWISE (:396-403, 1068-1086): temb [B, S_img, D] = one conditioning row per image token (the image stream's
    shift / scale / gate rows are then per token) with temb_txt [B, D] = its mean over the tokens for the text stream."""
    p = f"transformer_blocks.{i}."
    H = cfg.num_attention_heads
    m = linear(F.silu(temb), P, p + "norm1.linear", lora, lora_scale)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (_rows(t) for t in m.chunk(6, dim=-1))
    c = linear(F.silu(temb if temb_txt is None else temb_txt), P, p + "norm1_context.linear", lora, lora_scale)
    c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = c.chunk(6, dim=1)
    n = layer_norm(hidden) * (1 + scale_msa) + shift_msa
    cn = layer_norm(enc) * (1 + c_scale_msa[:, None]) + c_shift_msa[:, None]

    a = p + "attn."
    q = rms_norm(_heads(linear(n, P, a + "to_q", lora, lora_scale), H), P[a + "norm_q.weight"])
    k = rms_norm(_heads(linear(n, P, a + "to_k", lora, lora_scale), H), P[a + "norm_k.weight"])
    v = _heads(linear(n, P, a + "to_v", lora, lora_scale), H)
    cq = rms_norm(_heads(linear(cn, P, a + "add_q_proj", lora, lora_scale), H), P[a + "norm_added_q.weight"])
    ck = rms_norm(_heads(linear(cn, P, a + "add_k_proj", lora, lora_scale), H), P[a + "norm_added_k.weight"])
    cv = _heads(linear(cn, P, a + "add_v_proj", lora, lora_scale), H)
    q = torch.cat([cq, q], dim=2); k = torch.cat([ck, k], dim=2); v = torch.cat([cv, v], dim=2)   # [txt || img]
    q = apply_rope(q, cos, sin); k = apply_rope(k, cos, sin)
    o = sdpa(q, k, v, key_bias)
    B, _, S, _ = o.shape
    o = o.transpose(1, 2).reshape(B, S, -1)
    T = enc.shape[1]
    co, io = o[:, :T], o[:, T:]
    if taps is not None:
        taps[f"d{i}.attn"] = o
    hidden = hidden + gate_msa * linear(io, P, a + "to_out.0", lora, lora_scale)
    enc = enc + c_gate_msa[:, None] * linear(co, P, a + "to_add_out", lora, lora_scale)

    n2 = layer_norm(hidden) * (1 + scale_mlp) + shift_mlp
    ff = linear(F.gelu(linear(n2, P, p + "ff.net.0.proj", lora, lora_scale), approximate="tanh"), P, p + "ff.net.2", lora, lora_scale)
    hidden = hidden + gate_mlp * ff
    cn2 = layer_norm(enc) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
    cff = linear(F.gelu(linear(cn2, P, p + "ff_context.net.0.proj", lora, lora_scale), approximate="tanh"), P, p + "ff_context.net.2", lora, lora_scale)
    enc = enc + c_gate_mlp[:, None] * cff
    enc = torch.nan_to_num(enc, nan=0.0, posinf=65504, neginf=-65504)
    return enc, hidden


def single_block(P, cfg, i, x, temb, cos, sin, lora=None, lora_scale=1.0, key_bias=None):
    p = f"single_transformer_blocks.{i}."
    H = cfg.num_attention_heads
    m = linear(F.silu(temb), P, p + "norm.linear", lora, lora_scale)          # temb [B, D], or tokenwise [B, S_txt + S_img, D] (`temb_single`, :1075-1083)
    shift, scale, gate = (_rows(t) for t in m.chunk(3, dim=-1))
    n = layer_norm(x) * (1 + scale) + shift
    a = p + "attn."
    q = rms_norm(_heads(linear(n, P, a + "to_q", lora, lora_scale), H), P[a + "norm_q.weight"])
    k = rms_norm(_heads(linear(n, P, a + "to_k", lora, lora_scale), H), P[a + "norm_k.weight"])
    v = _heads(linear(n, P, a + "to_v", lora, lora_scale), H)
    q = apply_rope(q, cos, sin); k = apply_rope(k, cos, sin)
    o = sdpa(q, k, v, key_bias)
    B, _, S, _ = o.shape
    o = o.transpose(1, 2).reshape(B, S, -1)
    mlp = F.gelu(linear(n, P, p + "proj_mlp", lora, lora_scale), approximate="tanh")
    out = x + gate * linear(torch.cat([o, mlp], dim=2), P, p + "proj_out", lora, lora_scale)
    return torch.nan_to_num(out, nan=0.0, posinf=65504, neginf=-65504)


def flux_forward(P, cfg: FluxConfig, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                 guidance=None, lora=None, lora_scale: float = 1.0, key_bias=None, taps=None, tread=None, checkpoint: bool = False):
    """flux/transformer.py:940-1513.  hidden_states [B,S_img,64] packed latents; timestep in [0,1] (multiplied by 1000 here,
    :1003); guidance likewise (:1007).  Returns [B,S_img,64].
    tread = {"routes": [{start_layer_idx, end_layer_idx, ...}], "mask_infos": [{ids_shuffle, ids_restore, ids_keep}, ...]}: TREAD routing
    (:1095-1241 double blocks, :1394-1486 single blocks) with the router's permutations replayed; layer indices are global over
    double + single blocks, negative = from the end (:1120-1133).
    checkpoint=True re-runs each block in the backward (torch.utils.checkpoint, the reference's gradient_checkpointing branch :1137-1160): same values,
    bounded activation memory — what lets the full 19 + 38 block depth run as an fp32 checker."""
    if checkpoint:
        import torch.utils.checkpoint as _ck
        run = lambda fn, *a: _ck.checkpoint(fn, *a, use_reentrant=False)
    else:
        run = lambda fn, *a: fn(*a)
    hidden = linear(hidden_states, P, "x_embedder", lora, lora_scale)
    t = timestep.float() * 1000
    g = guidance.float() * 1000 if (guidance is not None and cfg.guidance_embeds) else None
    enc = linear(encoder_hidden_states, P, "context_embedder")
    temb_txt = temb_single = None
    if timestep.ndim == 2:
        # TOKENWISE timesteps [B, S_img] (:245-294 `_flux_tokenwise_conditioning`, :1068-1086): one conditioning row per image token; the text tokens take the
        # mean over the image tokens; the single blocks see [mean x S_txt || per token] along their joint sequence; norm_out takes the per-token rows (:1505)
        Bq, Sq = timestep.shape
        if Sq != hidden.shape[1]:
            raise ValueError(f"Flux expected tokenwise timesteps with sequence length {hidden.shape[1]}, got {Sq}.")
        if tread:
            raise ValueError("tokenwise timesteps under TREAD routing are not restated")
        gg = None
        if g is not None:
            gg = (g.expand(Bq) if g.numel() == 1 else g)
            gg = gg[:, None].expand(Bq, Sq).reshape(-1) if gg.ndim == 1 else gg.reshape(-1)
        temb = time_text_embed(P, cfg, t.reshape(-1), gg, pooled_projections[:, None, :].expand(-1, Sq, -1).reshape(Bq * Sq, -1)).view(Bq, Sq, -1)
        temb_txt = temb.mean(dim=1)
        temb_single = torch.cat([temb_txt[:, None].expand(-1, enc.shape[1], -1), temb], dim=1)
    else:
        temb = time_text_embed(P, cfg, t, g, pooled_projections)
    ids = torch.cat((txt_ids.cpu(), img_ids.cpu()), dim=0)
    cos, sin = (t.to(hidden.device) for t in rope_tables(ids, cfg.axes_dims_rope))      # tables in float64 on the host; the oracle itself may run on any device
    if taps is not None:
        taps["temb"] = temb; taps["x_embed"] = hidden; taps["ctx_embed"] = enc
    total = cfg.num_layers + cfg.num_single_layers
    routes = [dict(r, start_layer_idx=r["start_layer_idx"] % total, end_layer_idx=r["end_layer_idx"] % total) for r in (tread or {}).get("routes", [])]
    infos = (tread or {}).get("mask_infos", [])
    ptr, info, saved, gidx = 0, None, None, 0
    ccos, csin = cos, sin
    B, T = hidden.shape[0], enc.shape[1]
    for i in range(cfg.num_layers):
        if ptr < len(routes) and gidx == routes[ptr]["start_layer_idx"]:
            info, saved = infos[ptr], hidden
            hidden = tread_start(hidden, info)
            ccos, csin = tread_rope(cos, sin, T, info, hidden.shape[1], B)
        enc, hidden = run(double_block, P, cfg, i, hidden, enc, temb, ccos, csin, lora, lora_scale, key_bias, taps, temb_txt)
        if info is not None and gidx == routes[ptr]["end_layer_idx"]:
            hidden = tread_end(hidden, info, saved)
            info, saved, ptr, ccos, csin = None, None, ptr + 1, cos, sin
        if taps is not None:
            taps[f"d{i}.img"] = hidden; taps[f"d{i}.txt"] = enc
        gidx += 1
    x = torch.cat([enc, hidden], dim=1)
    for i in range(cfg.num_single_layers):
        if ptr < len(routes) and gidx == routes[ptr]["start_layer_idx"]:
            info, saved = infos[ptr], x[:, T:]
            img = tread_start(x[:, T:], info)
            x = torch.cat([x[:, :T], img], dim=1)
            ccos, csin = tread_rope(cos, sin, T, info, img.shape[1], B)
        x = run(single_block, P, cfg, i, x, temb if temb_single is None else temb_single, ccos, csin, lora, lora_scale, key_bias)
        if info is not None and gidx == routes[ptr]["end_layer_idx"]:
            x = torch.cat([x[:, :T], tread_end(x[:, T:], info, saved)], dim=1)
            info, saved, ptr, ccos, csin = None, None, ptr + 1, cos, sin
        if taps is not None:
            taps[f"s{i}"] = x
        gidx += 1
    hidden = x[:, T:]
    emb = linear(F.silu(temb), P, "norm_out.linear")
    scale, shift = emb.chunk(2, dim=-1)         # AdaLayerNormContinuous: scale FIRST
    hidden = layer_norm(hidden) * (1 + _rows(scale)) + _rows(shift)
    return linear(hidden, P, "proj_out", lora, lora_scale)


def flux_model_predict(P, cfg, noisy_latents, prompt_embeds, pooled, timesteps, guidance_value: float = 1.0, lora=None,
                       lora_scale=1.0, taps=None, checkpoint: bool = False, tread=None):
    """Flux._model_predict_single (flux/model.py:707-864): pack, ids, t/1000, guidance vector, transformer, unpack."""
    B, C, Hh, Ww = noisy_latents.shape
    packed = pack_latents(noisy_latents)
    img_ids = prepare_latent_image_ids(Hh, Ww)
    txt_ids = torch.zeros(prompt_embeds.shape[1], 3)
    guidance = torch.full((B,), float(guidance_value), device=noisy_latents.device) if cfg.guidance_embeds else None
    out = flux_forward(P, cfg, packed, prompt_embeds, pooled, timesteps / 1000.0, img_ids, txt_ids, guidance, lora, lora_scale,
                       taps=taps, checkpoint=checkpoint, tread=tread)
    return unpack_latents(out, Hh, Ww)


def train_flops_per_image(cfg: FluxConfig, S_img: int, S_txt: int, lora: bool = True) -> float:
    """per block 2*S*12D^2 (linears) + 4*S^2*D (attention) forward; LoRA step = 2x linears + 3x attention."""
    D = cfg.inner_dim
    S = S_img + S_txt
    nblk = cfg.num_layers + cfg.num_single_layers
    lin = nblk * 2.0 * S * 12 * D * D
    att = nblk * 4.0 * S * S * D
    return (2 * lin + 3 * att) if lora else 3 * (lin + att)
