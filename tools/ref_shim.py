#!/usr/bin/env python3
"""Make the reference's own model files IMPORTABLE in this container, so their control flow can be EXECUTED (not restated).

TEST INFRASTRUCTURE (fixture generation only; nothing under simpletuner_amd/ or tests/ imports this at run time — the GPU box has no
/root/reference).  Used by tools/gen_ref_models.py.

Why a shim: every hot-path module of the reference does `from diffusers ...` at import time, and diffusers / peft are not installed
(SURVEY.md F2/F3); the `simpletuner` package cannot be imported as a package either (its `__init__` chain eagerly imports 40 model
families, python >= 3.12).  `install()` therefore

  1. registers EMPTY package objects for `simpletuner`, `simpletuner.helpers`, ... whose `__path__` points at the real directories
     under /root/reference, so `import simpletuner.helpers.models.flux.transformer` loads THAT FILE unmodified while no `__init__.py`
     of the reference ever runs;
  2. registers tiny stand-ins for the reference's heavy side modules the model files import but the default path never calls
     (attention_backend's Metal fast path, the CPU-offloading checkpointer, QK-clip logging, GLIGEN layers);
  3. registers a minimal fake `diffusers` package with the LEAF modules the reference's files construct: Attention (+ the
     JointAttnProcessor2_0 / AttnProcessor2_0 processors), FeedForward, AdaLayerNormZero / -ZeroSingle / -Continuous / -Single,
     SD35AdaLayerNormZeroX, JointTransformerBlock (constructor only — its arithmetic is the reference's
     `_sd3_apply_joint_transformer_block`), BasicTransformerBlock, PatchEmbed, the timestep / text-projection embedders,
     FluxPosEmbed, ConfigMixin / ModelMixin.  Each leaf follows the public diffusers (>= 0.36) definition; where the reference
     tree VENDORS the same leaf for another model family, that vendored code is lifted by AST and used instead of a restatement
     (RMSNorm, Timesteps, TimestepEmbedding: helpers/models/heartmula/codec/transformer.py:15-25, 410-440;
     AdaLayerNormContinuous's constructor: helpers/models/mageflow/vendor/models/modules/mage_layers.py:717-754; the 1-D rotary table of
     FluxPosEmbed is cross-checked against helpers/models/hunyuanvideo/modules/posemb_layers.py:275-323 and the sinusoid against
     helpers/models/qwen_image/transformer.py:237-290).

What this pins and what it does not: everything ABOVE the leaves — block wiring, chunk orders, gating, concatenation orders, mask
handling, residual placement, the model-level forward incl. its activation-checkpoint plans and TREAD routing — is the reference's
code, executed.  The leaves themselves (a Linear + LayerNorm + SiLU each) remain restatements of diffusers unless marked "lifted".
"""
from __future__ import annotations

import ast
import contextlib
import importlib
import inspect
import math
import sys
import types
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = Path("/root/reference")
REF = REF_ROOT / "simpletuner"


# ------------------------------------------------------------------------------------------------------------------------
# AST lifting of vendored leaves
# ------------------------------------------------------------------------------------------------------------------------
def lift(path: Path, names, extra_ns=None):
    """compile the named top-level classes / functions of a reference file in isolation (never copied into this repo)"""
    tree = ast.parse(path.read_text())
    picked = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise KeyError(f"{path}: missing {missing}")
    for n in picked:
        n.decorator_list = []
    mod = ast.Module(body=picked, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"torch": torch, "nn": nn, "F": F, "math": math, "Optional": Optional, "Tuple": Tuple, "Union": Any, "Any": Any,
          "__builtins__": __builtins__}
    ns.update(extra_ns or {})
    exec(compile(mod, str(path), "exec"), ns)
    return [ns[n] for n in names]


# ------------------------------------------------------------------------------------------------------------------------
# fake diffusers: plumbing classes
# ------------------------------------------------------------------------------------------------------------------------
class _Config(dict):
    """diffusers FrozenDict: a mapping with attribute access"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    def register_to_config(self, **kwargs):
        cfg = dict(getattr(self, "_internal_dict", {}))
        cfg.update(kwargs)
        object.__setattr__(self, "_internal_dict", _Config(cfg))

    @property
    def config(self):
        return self._internal_dict


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class PeftAdapterMixin:
    pass


class FromOriginalModelMixin:
    pass


class _CP:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Transformer2DModelOutput:
    def __init__(self, sample):
        self.sample = sample


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


class _logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()


def is_torch_version(op, ver):
    from packaging import version
    import operator as _op

    ops = {">=": _op.ge, ">": _op.gt, "<=": _op.le, "<": _op.lt, "==": _op.eq}
    return ops[op](version.parse(torch.__version__.split("+")[0]), version.parse(ver))


def _identity_decorator(x):
    return x


# ------------------------------------------------------------------------------------------------------------------------
# leaves
# ------------------------------------------------------------------------------------------------------------------------
RMSNorm, TimestepEmbedding, Timesteps = lift(REF / "helpers/models/heartmula/codec/transformer.py", ["RMSNorm", "TimestepEmbedding", "Timesteps"])
(get_timestep_embedding,) = lift(REF / "helpers/models/qwen_image/transformer.py", ["get_timestep_embedding"])
(get_1d_rotary_pos_embed_vendored,) = lift(REF / "helpers/models/hunyuanvideo/modules/posemb_layers.py", ["get_1d_rotary_pos_embed"])
(_VendoredAdaLNContinuous,) = lift(REF / "helpers/models/mageflow/vendor/models/modules/mage_layers.py", ["AdaLayerNormContinuous"],
                                   extra_ns={"RMSNorm": RMSNorm, "Tensor": torch.Tensor})


class AdaLayerNormContinuous(_VendoredAdaLNContinuous):
    """diffusers AdaLayerNormContinuous.  Constructor = the in-tree vendored class (mage_layers.py:717-754: SiLU, Linear(cond -> 2D), LayerNorm);
    forward restated for the batch-wise [B, D] conditioning diffusers takes (the vendored forward is a packed-sequence variant): chunk order
    scale, shift and the broadcast over tokens as in the reference's own tokenwise branch (flux/transformer.py:406-412, sd3/transformer.py:136-142)."""

    def forward(self, x, conditioning_embedding):
        emb = self.linear(self.silu(conditioning_embedding).to(x.dtype))
        scale, shift = torch.chunk(emb, 2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class _DiffusersRMSNorm(RMSNorm):
    """diffusers RMSNorm(dim, eps, elementwise_affine=True): the vendored class above with diffusers' constructor order"""

    def __init__(self, dim, eps: float = 1e-6, elementwise_affine: bool = True):
        super().__init__(dim, eps)
        assert elementwise_affine


class GELU(nn.Module):
    """diffusers.models.activations.GELU: Linear then gelu(approximate)"""

    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class GEGLU(nn.Module):
    """diffusers.models.activations.GEGLU: one Linear to 2 x dim_out, hidden * gelu(gate) (exact gelu) — the UNet's feed-forward activation"""

    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, x):
        hidden, gate = self.proj(x).chunk(2, dim=-1)
        return hidden * F.gelu(gate)


class FeedForward(nn.Module):
    """diffusers.models.attention.FeedForward (activation_fn in {"gelu", "gelu-approximate", "geglu"}): net = [act(proj), Dropout, Linear]"""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False, inner_dim=None, bias=True):
        super().__init__()
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        dim_out = dim if dim_out is None else dim_out
        if activation_fn == "gelu":
            act = GELU(dim, inner_dim, bias=bias)
        elif activation_fn == "gelu-approximate":
            act = GELU(dim, inner_dim, approximate="tanh", bias=bias)
        elif activation_fn == "geglu":
            act = GEGLU(dim, inner_dim, bias=bias)
        else:
            raise NotImplementedError(activation_fn)
        self.net = nn.ModuleList([act, nn.Dropout(dropout), nn.Linear(inner_dim, dim_out, bias=bias)])

    def forward(self, x, *a, **k):
        for m in self.net:
            x = m(x)
        return x


def _chunked_feed_forward(ff, hidden_states, chunk_dim, chunk_size):
    n = hidden_states.shape[chunk_dim] // chunk_size
    return torch.cat([ff(h) for h in hidden_states.chunk(n, dim=chunk_dim)], dim=chunk_dim)


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features, hidden_size, out_features=None, act_fn="gelu_tanh"):
        super().__init__()
        out_features = hidden_size if out_features is None else out_features
        self.linear_1 = nn.Linear(in_features, hidden_size, bias=True)
        self.act_1 = {"gelu_tanh": nn.GELU(approximate="tanh"), "silu": nn.SiLU()}[act_fn]
        self.linear_2 = nn.Linear(hidden_size, out_features, bias=True)

    def forward(self, caption):
        return self.linear_2(self.act_1(self.linear_1(caption)))


class CombinedTimestepTextProjEmbeddings(nn.Module):
    def __init__(self, embedding_dim, pooled_projection_dim):
        super().__init__()
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=embedding_dim)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, act_fn="silu")

    def forward(self, timestep, pooled_projection):
        t = self.timestep_embedder(self.time_proj(timestep).to(dtype=pooled_projection.dtype))
        return t + self.text_embedder(pooled_projection)


class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    def __init__(self, embedding_dim, pooled_projection_dim):
        super().__init__()
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=embedding_dim)
        self.guidance_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=embedding_dim)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, act_fn="silu")

    def forward(self, timestep, guidance, pooled_projection):
        t = self.timestep_embedder(self.time_proj(timestep).to(dtype=pooled_projection.dtype))
        g = self.guidance_embedder(self.time_proj(guidance).to(dtype=pooled_projection.dtype))
        return t + g + self.text_embedder(pooled_projection)


class PixArtAlphaCombinedTimestepSizeEmbeddings(nn.Module):
    def __init__(self, embedding_dim, size_emb_dim, use_additional_conditions=False):
        super().__init__()
        self.outdim = size_emb_dim
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=embedding_dim)
        self.use_additional_conditions = use_additional_conditions
        if use_additional_conditions:
            self.additional_condition_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
            self.resolution_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=size_emb_dim)
            self.aspect_ratio_embedder = TimestepEmbedding(in_channels=256, time_embed_dim=size_emb_dim)

    def forward(self, timestep, resolution, aspect_ratio, batch_size, hidden_dtype):
        timesteps_emb = self.timestep_embedder(self.time_proj(timestep).to(dtype=hidden_dtype))
        if self.use_additional_conditions:
            r = self.resolution_embedder(self.additional_condition_proj(resolution.flatten()).to(hidden_dtype)).reshape(batch_size, -1)
            a = self.aspect_ratio_embedder(self.additional_condition_proj(aspect_ratio.flatten()).to(hidden_dtype)).reshape(batch_size, -1)
            return timesteps_emb + torch.cat([r, a], dim=1)
        return timesteps_emb


class AdaLayerNormSingle(nn.Module):
    """in-tree analogue: helpers/models/heartmula/codec/transformer.py:394-408 (AdaLayerNormSingleFlow)"""

    def __init__(self, embedding_dim, use_additional_conditions=False):
        super().__init__()
        self.emb = PixArtAlphaCombinedTimestepSizeEmbeddings(embedding_dim, size_emb_dim=embedding_dim // 3,
                                                             use_additional_conditions=use_additional_conditions)
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 6 * embedding_dim, bias=True)

    def forward(self, timestep, added_cond_kwargs=None, batch_size=None, hidden_dtype=None):
        added_cond_kwargs = added_cond_kwargs or {"resolution": None, "aspect_ratio": None}
        embedded = self.emb(timestep, **added_cond_kwargs, batch_size=batch_size, hidden_dtype=hidden_dtype)
        return self.linear(self.silu(embedded)), embedded


class AdaLayerNormZero(nn.Module):
    """chunk order shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp (= the in-tree tokenwise variants
    flux/transformer.py:396-403, sd3/transformer.py:126-133)"""

    def __init__(self, embedding_dim, num_embeddings=None, norm_type="layer_norm", bias=True):
        super().__init__()
        assert num_embeddings is None and norm_type == "layer_norm"
        self.emb = None
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 6 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, timestep=None, class_labels=None, hidden_dtype=None, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormZeroSingle(nn.Module):
    def __init__(self, embedding_dim, norm_type="layer_norm", bias=True):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 3 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa = emb.chunk(3, dim=1)
        return self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None], gate_msa


class SD35AdaLayerNormZeroX(nn.Module):
    """SD3.5 dual-attention norm1: 9 chunks; returns (x, gate_msa, shift_mlp, scale_mlp, gate_mlp, x2, gate_msa2) as the reference
    unpacks it (sd3/transformer.py:155-165)"""

    def __init__(self, embedding_dim, norm_type="layer_norm", bias=True):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 9 * embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, elementwise_affine=False, eps=1e-6)

    def forward(self, hidden_states, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp, shift_msa2, scale_msa2, gate_msa2 = emb.chunk(9, dim=1)
        n = self.norm(hidden_states)
        return (n * (1 + scale_msa[:, None]) + shift_msa[:, None], gate_msa, shift_mlp, scale_mlp, gate_mlp,
                n * (1 + scale_msa2[:, None]) + shift_msa2[:, None], gate_msa2)


# --- attention -----------------------------------------------------------------------------------------------------------
class AttnProcessor2_0:
    """diffusers AttnProcessor2_0 (self / cross attention through F.scaled_dot_product_attention)"""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *args, **kwargs):
        residual = hidden_states
        batch_size, sequence_length, _ = hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape
        if attention_mask is not None:
            attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)
            attention_mask = attention_mask.view(batch_size, attn.heads, -1, attention_mask.shape[-1])
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        head_dim = key.shape[-1] // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        if attn.norm_q is not None:
            query = attn.norm_q(query)
        if attn.norm_k is not None:
            key = attn.norm_k(key)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim).to(query.dtype)
        hidden_states = attn.to_out[1](attn.to_out[0](hidden_states))
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        return hidden_states / attn.rescale_output_factor


AttnProcessor = AttnProcessor2_0
FusedAttnProcessor2_0 = AttnProcessor2_0          # only named by packed_attention_processors.py's import line


class JointAttnProcessor2_0:
    """diffusers JointAttnProcessor2_0: joint sequence = [sample || context]; corroborated in-tree by the packed processor's ordering
    (helpers/training/packed_attention_processors.py:159-174)"""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, *args, **kwargs):
        residual = hidden_states
        batch_size = hidden_states.shape[0]
        query, key, value = attn.to_q(hidden_states), attn.to_k(hidden_states), attn.to_v(hidden_states)
        head_dim = key.shape[-1] // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        if attn.norm_q is not None:
            query = attn.norm_q(query)
        if attn.norm_k is not None:
            key = attn.norm_k(key)
        if encoder_hidden_states is not None:
            eq = attn.add_q_proj(encoder_hidden_states).view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
            ek = attn.add_k_proj(encoder_hidden_states).view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
            ev = attn.add_v_proj(encoder_hidden_states).view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
            if attn.norm_added_q is not None:
                eq = attn.norm_added_q(eq)
            if attn.norm_added_k is not None:
                ek = attn.norm_added_k(ek)
            query = torch.cat([query, eq], dim=2)
            key = torch.cat([key, ek], dim=2)
            value = torch.cat([value, ev], dim=2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, dropout_p=0.0, is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim).to(query.dtype)
        if encoder_hidden_states is not None:
            hidden_states, encoder_hidden_states = hidden_states[:, : residual.shape[1]], hidden_states[:, residual.shape[1]:]
            if not attn.context_pre_only:
                encoder_hidden_states = attn.to_add_out(encoder_hidden_states)
        hidden_states = attn.to_out[1](attn.to_out[0](hidden_states))
        if encoder_hidden_states is not None:
            return hidden_states, encoder_hidden_states
        return hidden_states


class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention: the projections, the optional per-head q/k norms and the processor hook.
    Usage corroborated in-tree: flux/transformer.py:127-148, 218-221, 440-451, 539-551."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, kv_heads=None, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None, cross_attention_norm_num_groups=32,
                 qk_norm=None, added_kv_proj_dim=None, added_proj_bias=True, norm_num_groups=None, spatial_norm_dim=None,
                 out_bias=True, scale_qk=True, only_cross_attention=False, eps=1e-5, rescale_output_factor=1.0,
                 residual_connection=False, _from_deprecated_attn_block=False, processor=None, out_dim=None, out_context_dim=None,
                 context_pre_only=None, pre_only=False, elementwise_affine=True, is_causal=False):
        super().__init__()
        self.inner_dim = out_dim if out_dim is not None else dim_head * heads
        self.query_dim = query_dim
        self.is_cross_attention = cross_attention_dim is not None
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.out_dim = out_dim if out_dim is not None else query_dim
        self.out_context_dim = out_context_dim if out_context_dim is not None else query_dim
        self.context_pre_only = context_pre_only
        self.pre_only = pre_only
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = out_dim // dim_head if out_dim is not None else heads
        self.added_kv_proj_dim = added_kv_proj_dim
        self.only_cross_attention = only_cross_attention
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        if qk_norm is None:
            self.norm_q = self.norm_k = None
        elif qk_norm == "rms_norm":
            self.norm_q, self.norm_k = _DiffusersRMSNorm(dim_head, eps=eps), _DiffusersRMSNorm(dim_head, eps=eps)
        else:
            raise NotImplementedError(qk_norm)
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        if added_kv_proj_dim is not None:
            self.add_k_proj = nn.Linear(added_kv_proj_dim, self.inner_dim, bias=added_proj_bias)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, self.inner_dim, bias=added_proj_bias)
            if context_pre_only is not None:
                self.add_q_proj = nn.Linear(added_kv_proj_dim, self.inner_dim, bias=added_proj_bias)
        else:
            self.add_q_proj = self.add_k_proj = self.add_v_proj = None
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, self.out_dim, bias=out_bias), nn.Dropout(dropout)])
        else:
            self.to_out = None
        if context_pre_only is not None and not context_pre_only:
            self.to_add_out = nn.Linear(self.inner_dim, self.out_context_dim, bias=out_bias)
        else:
            self.to_add_out = None
        if qk_norm is not None and added_kv_proj_dim is not None:
            self.norm_added_q, self.norm_added_k = _DiffusersRMSNorm(dim_head, eps=eps), _DiffusersRMSNorm(dim_head, eps=eps)
        else:
            self.norm_added_q = self.norm_added_k = None
        self.processor = processor if processor is not None else AttnProcessor2_0()

    def set_processor(self, processor):
        self.processor = processor

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        head_size = self.heads
        if attention_mask is None:
            return attention_mask
        if attention_mask.shape[-1] != target_length:
            attention_mask = F.pad(attention_mask, (0, target_length), value=0.0)
        if out_dim == 3:
            if attention_mask.shape[0] < batch_size * head_size:
                attention_mask = attention_mask.repeat_interleave(head_size, dim=0)
        elif out_dim == 4:
            attention_mask = attention_mask.unsqueeze(1).repeat_interleave(head_size, dim=1)
        return attention_mask

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        params = set(inspect.signature(self.processor.__call__).parameters.keys())
        has_var = any(p.kind == inspect.Parameter.VAR_KEYWORD for p in inspect.signature(self.processor.__call__).parameters.values())
        kw = {k: v for k, v in cross_attention_kwargs.items() if has_var or k in params}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)


# --- blocks whose ARITHMETIC the reference carries in-tree ---------------------------------------------------------------------
class JointTransformerBlock(nn.Module):
    """diffusers JointTransformerBlock — CONSTRUCTOR ONLY (structure corroborated by helpers/models/sd3/expanded.py:31-110).  The
    reference never calls its forward: `_sd3_apply_joint_transformer_block` (sd3/transformer.py:145-241) is the arithmetic."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, context_pre_only=False, qk_norm=None, use_dual_attention=False):
        super().__init__()
        self.use_dual_attention = use_dual_attention
        self.context_pre_only = context_pre_only
        self.norm1 = SD35AdaLayerNormZeroX(dim) if use_dual_attention else AdaLayerNormZero(dim)
        if context_pre_only:
            self.norm1_context = AdaLayerNormContinuous(dim, dim, elementwise_affine=False, eps=1e-6, bias=True, norm_type="layer_norm")
        else:
            self.norm1_context = AdaLayerNormZero(dim)
        self.attn = Attention(query_dim=dim, cross_attention_dim=None, added_kv_proj_dim=dim, dim_head=attention_head_dim,
                              heads=num_attention_heads, out_dim=dim, context_pre_only=context_pre_only, bias=True,
                              processor=JointAttnProcessor2_0(), qk_norm=qk_norm, eps=1e-6)
        if use_dual_attention:
            self.attn2 = Attention(query_dim=dim, cross_attention_dim=None, dim_head=attention_head_dim, heads=num_attention_heads,
                                   out_dim=dim, bias=True, processor=JointAttnProcessor2_0(), qk_norm=qk_norm, eps=1e-6)
        else:
            self.attn2 = None
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate")
        if not context_pre_only:
            self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
            self.ff_context = FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate")
        else:
            self.norm2_context = None
            self.ff_context = None
        self._chunk_size = None
        self._chunk_dim = 0

    def forward(self, *a, **k):
        raise RuntimeError("the reference drives JointTransformerBlock through _sd3_apply_joint_transformer_block")


class BasicTransformerBlock(nn.Module):
    """diffusers BasicTransformerBlock, `ada_norm_single` form (PixArt).  The batch-wise forward below restates diffusers; the same
    arithmetic exists in-tree as the tokenwise override PixArtSelfFlowTransformerBlock.forward (pixart/transformer.py:95-145), which
    tools/gen_ref_models.py ALSO executes on the same inputs (a [B, S, 6D] broadcast of the [B, 6D] modulation) and requires to agree."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, cross_attention_dim=None, activation_fn="geglu",
                 num_embeds_ada_norm=None, attention_bias=False, only_cross_attention=False, double_self_attention=False,
                 upcast_attention=False, norm_elementwise_affine=True, norm_type="layer_norm", norm_eps=1e-5, final_dropout=False,
                 attention_type="default", positional_embeddings=None, num_positional_embeddings=None, ff_inner_dim=None,
                 ff_bias=True, attention_out_bias=True):
        super().__init__()
        assert norm_type in ("ada_norm_single", "layer_norm") and positional_embeddings is None
        self.dim = dim
        self.norm_type = norm_type
        self.only_cross_attention = only_cross_attention
        self.pos_embed = None
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=norm_elementwise_affine, eps=norm_eps)
        self.attn1 = Attention(query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout, bias=attention_bias,
                               cross_attention_dim=cross_attention_dim if only_cross_attention else None, upcast_attention=upcast_attention,
                               out_bias=attention_out_bias)
        if cross_attention_dim is not None or double_self_attention:
            self.norm2 = nn.LayerNorm(dim, norm_eps, norm_elementwise_affine)
            self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim if not double_self_attention else None,
                                   heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout, bias=attention_bias,
                                   upcast_attention=upcast_attention, out_bias=attention_out_bias)
        else:
            self.norm2 = nn.LayerNorm(dim, norm_eps, norm_elementwise_affine)
            self.attn2 = None
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn, final_dropout=final_dropout, inner_dim=ff_inner_dim, bias=ff_bias)
        if norm_type == "ada_norm_single":
            self.scale_shift_table = nn.Parameter(torch.randn(6, dim) / dim ** 0.5)
        else:                                  # "layer_norm" (the conv UNets' Transformer2DModel): a third LayerNorm in front of the feed-forward
            self.norm3 = nn.LayerNorm(dim, norm_eps, norm_elementwise_affine)
        self._chunk_size = None
        self._chunk_dim = 0

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None, timestep=None,
                cross_attention_kwargs=None, class_labels=None, added_cond_kwargs=None):
        if self.norm_type == "layer_norm":     # diffusers BasicTransformerBlock.forward, norm_type == "layer_norm": three pre-norm residual branches
            attn_output = self.attn1(self.norm1(hidden_states), encoder_hidden_states=encoder_hidden_states if self.only_cross_attention else None,
                                     attention_mask=attention_mask)
            hidden_states = attn_output + hidden_states
            if self.attn2 is not None:
                attn_output = self.attn2(self.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states, attention_mask=encoder_attention_mask)
                hidden_states = attn_output + hidden_states
            return self.ff(self.norm3(hidden_states)) + hidden_states
        batch_size = hidden_states.shape[0]
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (
            self.scale_shift_table[None] + timestep.reshape(batch_size, 6, -1)).chunk(6, dim=1)
        norm_hidden_states = self.norm1(hidden_states) * (1 + scale_msa) + shift_msa
        attn_output = self.attn1(norm_hidden_states, encoder_hidden_states=encoder_hidden_states if self.only_cross_attention else None,
                                 attention_mask=attention_mask)
        hidden_states = gate_msa * attn_output + hidden_states
        if self.attn2 is not None:
            attn_output = self.attn2(hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=encoder_attention_mask)
            hidden_states = attn_output + hidden_states
        norm_hidden_states = self.norm2(hidden_states) * (1 + scale_mlp) + shift_mlp
        ff_output = self.ff(norm_hidden_states)
        return gate_mlp * ff_output + hidden_states


# --- embeddings -------------------------------------------------------------------------------------------------------------
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    omega = torch.arange(embed_dim // 2, dtype=torch.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = torch.outer(pos.reshape(-1).to(torch.float64), omega)
    return torch.cat([out.sin(), out.cos()], dim=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, interpolation_scale=1.0, base_size=16):
    """diffusers get_2d_sincos_pos_embed (output_type="pt"): w coordinate first, sin then cos per half"""
    if isinstance(grid_size, int):
        grid_size = (grid_size, grid_size)
    grid_h = torch.arange(grid_size[0], dtype=torch.float32) / (grid_size[0] / base_size) / interpolation_scale
    grid_w = torch.arange(grid_size[1], dtype=torch.float32) / (grid_size[1] / base_size) / interpolation_scale
    grid = torch.stack(torch.meshgrid(grid_w, grid_h, indexing="xy"), dim=0).reshape(2, 1, grid_size[1], grid_size[0])
    emb_h = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
    emb_w = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
    return torch.cat([emb_h, emb_w], dim=1)


class PatchEmbed(nn.Module):
    """diffusers PatchEmbed: Conv2d(k = s = patch) -> tokens + 2-D sincos table (centre crop of a pos_embed_max_size grid for SD3;
    the sample-size table, recomputed for other (h, w), for PixArt).  UNCORROBORATED in-tree (only call sites: sd3/transformer.py:329-336,
    pixart/transformer.py:305-312)."""

    def __init__(self, height=224, width=224, patch_size=16, in_channels=3, embed_dim=768, layer_norm=False, flatten=True, bias=True,
                 interpolation_scale=1, pos_embed_type="sincos", pos_embed_max_size=None):
        super().__init__()
        self.flatten, self.pos_embed_max_size, self.patch_size = flatten, pos_embed_max_size, patch_size
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=(patch_size, patch_size), stride=patch_size, bias=bias)
        self.norm = None
        self.height, self.width = height // patch_size, width // patch_size
        self.base_size = height // patch_size
        self.interpolation_scale = interpolation_scale
        grid_size = pos_embed_max_size if pos_embed_max_size else int(((height // patch_size) * (width // patch_size)) ** 0.5)
        pos_embed = get_2d_sincos_pos_embed(embed_dim, grid_size, base_size=self.base_size, interpolation_scale=self.interpolation_scale)
        self.register_buffer("pos_embed", pos_embed.float().unsqueeze(0), persistent=bool(pos_embed_max_size))

    def cropped_pos_embed(self, height, width):
        height, width = height // self.patch_size, width // self.patch_size
        top, left = (self.pos_embed_max_size - height) // 2, (self.pos_embed_max_size - width) // 2
        sp = self.pos_embed.reshape(1, self.pos_embed_max_size, self.pos_embed_max_size, -1)
        return sp[:, top:top + height, left:left + width, :].reshape(1, -1, sp.shape[-1])

    def forward(self, latent):
        if self.pos_embed_max_size is not None:
            height, width = latent.shape[-2:]
        else:
            height, width = latent.shape[-2] // self.patch_size, latent.shape[-1] // self.patch_size
        latent = self.proj(latent).flatten(2).transpose(1, 2)
        if self.pos_embed_max_size:
            pos_embed = self.cropped_pos_embed(height, width)
        elif self.height != height or self.width != width:
            pos_embed = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], (height, width), base_size=self.base_size,
                                                interpolation_scale=self.interpolation_scale).float().unsqueeze(0).to(latent.device)
        else:
            pos_embed = self.pos_embed
        return (latent + pos_embed).to(latent.dtype)


def get_1d_rotary_pos_embed(dim, pos, theta=10000.0, use_real=False, linear_factor=1.0, ntk_factor=1.0, repeat_interleave_real=True,
                            freqs_dtype=torch.float32):
    """diffusers get_1d_rotary_pos_embed (use_real, repeat_interleave_real): frequencies in `freqs_dtype`, outputs fp32"""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=freqs_dtype)[: dim // 2] / dim)) / linear_factor
    freqs = torch.outer(pos, freqs)
    return freqs.cos().repeat_interleave(2, dim=1).float(), freqs.sin().repeat_interleave(2, dim=1).float()


class FluxPosEmbed(nn.Module):
    """diffusers FluxPosEmbed: per-axis 1-D tables (float64 frequencies) concatenated -> (cos, sin) [S, sum(axes_dim)] fp32.
    The 1-D table is checked against the in-tree vendored function (hunyuanvideo/modules/posemb_layers.py:275-323) at install()."""

    def __init__(self, theta, axes_dim):
        super().__init__()
        self.theta, self.axes_dim = theta, axes_dim

    def forward(self, ids):
        pos = ids.float()
        cos_out, sin_out = [], []
        for i in range(ids.shape[-1]):
            cos, sin = get_1d_rotary_pos_embed(self.axes_dim[i], pos[:, i], theta=self.theta, use_real=True, repeat_interleave_real=True,
                                               freqs_dtype=torch.float64)
            cos_out.append(cos)
            sin_out.append(sin)
        return torch.cat(cos_out, dim=-1).to(ids.device), torch.cat(sin_out, dim=-1).to(ids.device)


def apply_rotary_emb(x, freqs_cis, use_real=True, use_real_unbind_dim=-1, sequence_dim=2):
    cos, sin = freqs_cis
    if sequence_dim == 2:
        cos, sin = cos[None, None], sin[None, None]
    else:
        cos, sin = cos[None, :, None], sin[None, :, None]
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos + rot.float() * sin).to(x.dtype)


# ------------------------------------------------------------------------------------------------------------------------
# install
# ------------------------------------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [str(path)]
    m.__package__ = name
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    # cross-checks of the leaves this file restates against in-tree vendored functions
    pos = torch.arange(0, 37, dtype=torch.float32)
    c0, s0 = get_1d_rotary_pos_embed(56, pos, use_real=True, freqs_dtype=torch.float64)
    c1, s1 = get_1d_rotary_pos_embed_vendored(56, pos, use_real=True)
    assert (c0 - c1).abs().max() < 2e-5 and (s0 - s1).abs().max() < 2e-5
    t = torch.tensor([0.0, 17.5, 999.0])
    assert torch.allclose(Timesteps(256, True, 0)(t), get_timestep_embedding(t, 256, flip_sin_to_cos=True, downscale_freq_shift=0), atol=1e-6)

    # 1. fake diffusers
    d = _mod("diffusers")
    d.__path__ = []
    for sub in ("models", "utils", "loaders", "configuration_utils", "models.attention", "models.attention_processor", "models.embeddings",
                "models.modeling_outputs", "models.modeling_utils", "models.normalization", "models.transformers",
                "models.transformers.transformer_flux", "models.transformers.transformer_2d", "models._modeling_parallel", "utils.torch_utils"):
        m = _mod("diffusers." + sub)
        m.__path__ = []
    S = sys.modules
    S["diffusers.configuration_utils"].__dict__.update(ConfigMixin=ConfigMixin, register_to_config=_identity_decorator)
    S["diffusers.loaders"].__dict__.update(FromOriginalModelMixin=FromOriginalModelMixin, PeftAdapterMixin=PeftAdapterMixin)
    S["diffusers.models._modeling_parallel"].__dict__.update(ContextParallelInput=_CP, ContextParallelOutput=_CP)
    S["diffusers.models.attention"].__dict__.update(FeedForward=FeedForward, _chunked_feed_forward=_chunked_feed_forward,
                                                    BasicTransformerBlock=BasicTransformerBlock, JointTransformerBlock=JointTransformerBlock)
    S["diffusers.models.attention_processor"].__dict__.update(Attention=Attention, AttentionProcessor=object, AttnProcessor=AttnProcessor,
                                                              AttnProcessor2_0=AttnProcessor2_0, JointAttnProcessor2_0=JointAttnProcessor2_0,
                                                              FusedAttnProcessor2_0=FusedAttnProcessor2_0)
    S["diffusers.models.embeddings"].__dict__.update(
        CombinedTimestepGuidanceTextProjEmbeddings=CombinedTimestepGuidanceTextProjEmbeddings,
        CombinedTimestepTextProjEmbeddings=CombinedTimestepTextProjEmbeddings, PatchEmbed=PatchEmbed,
        PixArtAlphaTextProjection=PixArtAlphaTextProjection, apply_rotary_emb=apply_rotary_emb, TimestepEmbedding=TimestepEmbedding,
        Timesteps=Timesteps)
    S["diffusers.models.modeling_outputs"].__dict__.update(Transformer2DModelOutput=Transformer2DModelOutput)
    S["diffusers.models.transformers.transformer_2d"].__dict__.update(Transformer2DModelOutput=Transformer2DModelOutput)
    S["diffusers.models.modeling_utils"].__dict__.update(ModelMixin=ModelMixin)
    S["diffusers.models.normalization"].__dict__.update(AdaLayerNormContinuous=AdaLayerNormContinuous, AdaLayerNormZero=AdaLayerNormZero,
                                                        AdaLayerNormZeroSingle=AdaLayerNormZeroSingle, AdaLayerNormSingle=AdaLayerNormSingle,
                                                        RMSNorm=_DiffusersRMSNorm, SD35AdaLayerNormZeroX=SD35AdaLayerNormZeroX)
    S["diffusers.models.transformers.transformer_flux"].__dict__.update(FluxPosEmbed=FluxPosEmbed)
    S["diffusers.utils"].__dict__.update(USE_PEFT_BACKEND=True, is_torch_version=is_torch_version, logging=_logging,
                                         scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None)
    S["diffusers.utils.torch_utils"].__dict__.update(maybe_allow_in_graph=_identity_decorator)
    d.FluxTransformer2DModel = object          # `from diffusers import FluxTransformer2DModel as Original...` (name only)
    S["diffusers.models"].PixArtTransformer2DModel = object      # type annotation in pixart/controlnet.py

    # 2. the reference tree as path-only packages (no __init__.py of the reference runs)
    _pkg("simpletuner", REF)
    for rel in ("helpers", "helpers/models", "helpers/models/flux", "helpers/models/sd3", "helpers/models/pixart", "helpers/training",
                "helpers/training/grounding", "helpers/utils"):
        _pkg("simpletuner." + rel.replace("/", "."), REF / rel)

    # 3. stand-ins for heavy side modules (features off on the default path)
    class _ABC:
        pass

    _mod("simpletuner.helpers.training.attention_backend", maybe_metal_flash_rope_attention=lambda *a, **k: None,
         get_packed_attention_backend=lambda *a, **k: None, AttentionBackendController=_ABC)
    _mod("simpletuner.helpers.training.offloaded_gradient_checkpointer",
         activation_offload_context=lambda enabled, label=None: contextlib.nullcontext(),
         offloaded_checkpoint=lambda fn, *a, **k: torch.utils.checkpoint.checkpoint(fn, *a, use_reentrant=False))
    _mod("simpletuner.helpers.training.qk_clip_logging", publish_attention_max_logits=lambda *a, **k: None)
    _mod("simpletuner.helpers.training.grounding.gligen_layers", apply_grounding_fuser=lambda fuser, hs, objs, **k: hs)


def ref_module(dotted: str):
    """import a reference module by its dotted name (after install())"""
    install()
    return importlib.import_module(dotted)
