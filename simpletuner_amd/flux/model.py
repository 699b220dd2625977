"""Flux model-family plugin — drop-in for simpletuner/helpers/models/flux/model.py on MI355X.

Same class attributes and step-path methods as the reference plugin (flux/model.py:49-114, 630-864):
`Flux(config, accelerator)`, `prepare_batch`, `model_predict -> {"model_prediction": [B,16,H,W]}`, `loss_with_logs`,
`get_trained_component`, `add_lora_adapter`.  `model_predict` divides the timesteps by 1000 in the batch dict exactly like
the reference (flux/model.py:739-745; pinned by tests/test_flux_model.py:213 there).
"""
from __future__ import annotations

import hashlib
import random

import torch

from .. import ops
from ..foundation import ModelFoundation, ModelRegistry, ModelTypes, PredictionTypes
from .transformer import FluxTransformer2DModel

BF16 = torch.bfloat16


def prepare_latent_image_ids(batch_size, height, width, device, dtype):
    """flux/__init__.py:48-63 — [ (H/2)(W/2), 3 ] fp32 ids (row, col in channels 1, 2)"""
    ids = torch.zeros(height // 2, width // 2, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(height // 2)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(width // 2)[None, :]
    return ids.reshape(-1, 3).to(device=device, dtype=torch.float32)


def pack_latents(latents, batch_size=None, num_channels_latents=None, height=None, width=None):
    """flux/__init__.py:25-31 (HIP permute kernel)"""
    return ops.flux_pack(latents)


def unpack_latents(latents, height, width, vae_scale_factor):
    """flux/__init__.py:34-45: height/width are PIXEL sizes, vae_scale_factor=16 -> token grid; returns [B, C/4, 2h, 2w]"""
    h, w = height // vae_scale_factor, width // vae_scale_factor
    return ops.flux_unpack(latents, latents.shape[-1] // 4, h * 2, w * 2)


class Flux(ModelFoundation):
    NAME = "Flux.1"
    PREDICTION_TYPE = PredictionTypes.FLOW_MATCHING
    MODEL_TYPE = ModelTypes.TRANSFORMER
    MODEL_CLASS = FluxTransformer2DModel
    MODEL_SUBFOLDER = "transformer"
    LATENT_CHANNEL_COUNT = 16
    COMFYUI_LORA_PRESERVE_COMPONENT_PREFIXES = {"transformer"}      # flux/model.py:56-57
    AUTO_LORA_FORMAT_DETECTION = True
    VAE_CONFIG = dict(latent_channels=16, scaling_factor=0.3611, shift_factor=0.1159, use_quant_conv=False)
    DEFAULT_MODEL_FLAVOUR = "dev"
    DEFAULT_LORA_TARGET = ["to_k", "to_q", "to_v", "to_out.0"]

    def convert_text_embed_for_pipeline(self, text_embedding: dict) -> dict:
        """flux/model.py:453-472: prompt / pooled embeddings (+ the text mask as `prompt_mask` only under flux_attention_masked_training)"""
        out = self._convert_text_embed(text_embedding, negative=False)
        m = text_embedding.get("attention_masks", None)
        if m is not None and m.dim() == 1:
            m = m.unsqueeze(0)
        out["prompt_mask"] = m if getattr(self.config, "flux_attention_masked_training", False) else None
        return out

    def convert_negative_text_embed_for_pipeline(self, text_embedding: dict) -> dict:
        """flux/model.py:474-497: no negative branch unless real CFG is on (validation_guidance_real > 1)"""
        g = getattr(self.config, "validation_guidance_real", None)
        if g is None or g <= 1.0:
            return {}
        out = self._convert_text_embed(text_embedding, negative=True)
        m = text_embedding.get("attention_masks", None)
        if m is not None and m.dim() == 1:
            m = m.unsqueeze(0)
        out["negative_mask"] = m if getattr(self.config, "flux_attention_masked_training", False) else None
        out["guidance_scale_real"] = float(g)
        out["no_cfg_until_timestep"] = int(getattr(self.config, "validation_no_cfg_until_timestep", 0) or 0)
        return out
    HUGGINGFACE_PATHS = {"dev": "black-forest-labs/flux.1-dev", "schnell": "black-forest-labs/flux.1-schnell"}

    def __init__(self, config, accelerator):
        super().__init__(config, accelerator)
        self._ids_cache = {}

    # ---- loading (common.py:3400 load_model).  Checkpoints are absent offline: synthetic init or a provided state dict ----
    def load_model(self, state_dict=None, **arch):
        self.model = FluxTransformer2DModel(device=self.accelerator.device, **arch)
        if state_dict is not None:
            self.model.load_flat_state(state_dict)
        else:
            self.model.init_synthetic(seed=int(getattr(self.config, "seed", 42) or 42))
        return self.model

    def add_lora_adapter(self):
        """common.py:1049-1128"""
        if getattr(self.config, "model_type", "lora") != "lora":
            raise RuntimeError("model_type == 'full' trains every transformer parameter: call enable_full_finetune() instead of add_lora_adapter()")
        targets = self._lora_target_set()
        params = self.unwrap_model(self.model).add_lora_adapter(rank=int(self.config.lora_rank),
                                                                alpha=getattr(self.config, "lora_alpha", None), targets=targets,
                                                                seed=int(getattr(self.config, "seed", 42) or 42) + 7,
                                                                init_b_std=float(getattr(self.config, "lora_init_b_std", 0.0)))
        self.unwrap_model(self.model).prepare_for_training()
        return params

    def enable_full_finetune(self):
        """model_type == "full" — the reference's multi-GPU Flux datapoint (documentation/DISTRIBUTED.md:291-298) trains the whole 12 B-parameter transformer:
        every weight, bias, q / k RMSNorm weight and modulation row (bf16 parameters + bf16 gradients in two arenas of one layout)"""
        return self.unwrap_model(self.model).enable_full_finetune()

    # flux/model.py:1235-1380: `flux_lora_target` names a set of wrapped Linears.  Built here: the attention projections — "all" (image + context
    # stream: to_q/k/v, add_q/k/v_proj, to_out.0, to_add_out; the reference's default), "context", the fall-through DEFAULT_LORA_TARGET (image stream and single
    # blocks only: what BASELINE.json's config names) — the feed-forward sets "all+ffs" / "context+ffs" (ff.net.*, ff_context.net.*, proj_mlp, proj_out) and
    # "all+ffs+embedder" (+ x_embedder), "ai-toolkit" (+ the AdaLN modulation Linears), "tiny" / "nano" (single_transformer_blocks.7(.20).proj_out).  "controlnet" wraps the
    # layers of a Flux ControlNet, a model this path does not build: refused, never silently narrowed.
    _UNBUILT_LORA_TARGETS = ("controlnet",)
    _BUILT_LORA_TARGETS = ("all", "context", "all+ffs", "context+ffs", "all+ffs+embedder", "ai-toolkit", "tiny", "nano")

    def _lora_target_set(self) -> str:
        want = str(getattr(self.config, "flux_lora_target", "default") or "default")
        if want in self._UNBUILT_LORA_TARGETS:
            raise NotImplementedError(f"flux_lora_target={want!r} is not implemented on the st355 path (built: {', '.join(repr(t) for t in self._BUILT_LORA_TARGETS)} "
                                      f"and the default attention set)")
        return want if want in self._BUILT_LORA_TARGETS else "default"

    def get_lora_target_layers(self):
        which = self._lora_target_set()
        attn_all = ["to_k", "to_q", "to_v", "to_qkv", "add_qkv_proj", "add_k_proj", "add_q_proj", "add_v_proj", "to_out.0", "to_add_out"]
        ctx = ["add_k_proj", "add_q_proj", "add_v_proj", "add_qkv_proj", "to_add_out"]                     # flux/model.py:1263-1271
        if which == "all":
            return attn_all
        if which == "context":
            return ctx
        if which == "context+ffs":                   # flux/model.py:1272-1282
            return ctx + ["ff_context.net.0.proj", "ff_context.net.2"]
        if which == "all+ffs":                       # flux/model.py:1283-1301
            return attn_all + ["ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2", "proj_mlp", "proj_out"]
        if which == "all+ffs+embedder":              # flux/model.py:1320-1339
            return ["x_embedder", "to_k", "to_q", "to_v", "to_qkv", "add_qkv_proj", "to_out.0", "add_k_proj", "add_q_proj", "add_v_proj", "to_add_out",
                    "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2", "proj_mlp", "proj_out"]
        if which == "ai-toolkit":                    # flux/model.py:1340-1362 (ostris' ai-toolkit layout)
            return ["to_q", "to_k", "to_qkv", "add_qkv_proj", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out", "ff.net.0.proj", "ff.net.2",
                    "ff_context.net.0.proj", "ff_context.net.2", "norm.linear", "norm1.linear", "norm1_context.linear", "proj_mlp", "proj_out"]
        if which == "tiny":                          # flux/model.py:1363-1369
            return ["single_transformer_blocks.7.proj_out", "single_transformer_blocks.20.proj_out"]
        if which == "nano":                          # flux/model.py:1370-1375
            return ["single_transformer_blocks.7.proj_out"]
        return list(self.DEFAULT_LORA_TARGET)

    def _flux_guidance_scales(self, prepared_batch, batch_size):
        """flux/model.py:682-705: one guidance value per sample — constant, or uniform in [flux_guidance_min, flux_guidance_max]; under XM the
        draws are made for the ORIGINAL samples and repeated for every noise candidate (candidate-major)"""
        import random
        mode = getattr(self.config, "flux_guidance_mode", "constant")
        if mode == "constant":
            return [float(getattr(self.config, "flux_guidance_value", 1.0))] * batch_size
        if mode == "random-range":
            lo, hi = self.config.flux_guidance_min, self.config.flux_guidance_max
            k, b0 = prepared_batch.get("xm_candidate_count"), prepared_batch.get("xm_original_batch_size")
            if k and b0:
                return [random.uniform(lo, hi) for _ in range(int(b0))] * int(k)
            return [random.uniform(lo, hi) for _ in range(batch_size)]
        raise ValueError(f"Unsupported Flux guidance mode: {mode!r}.")

    def _model_predict_single(self, prepared_batch: dict):
        """flux/model.py:707-864"""
        # [B] or tokenwise [B, S_img] timesteps (flux/model.py:560-600 `_normalize_timesteps`); Kontext's clean reference-image tokens ride along at t = 0 (:602-618, 762-778)
        self._require_per_sample_timesteps(prepared_batch, tokenwise_ok=True, conditioning_ok=True)
        lat = prepared_batch["latents"]
        B, Cc, Hh, Ww = lat.shape
        dev = self.accelerator.device
        packed = pack_latents(prepared_batch["noisy_latents"].to(device=dev, dtype=BF16))
        comp = self.get_trained_component()
        guidance = None
        if getattr(comp.config, "guidance_embeds", False):
            scales = tuple(self._flux_guidance_scales(prepared_batch, B))
            if getattr(self.config, "flux_guidance_mode", "constant") == "constant":
                gkey = ("guidance", scales)
                if gkey not in self._ids_cache:           # cached: a host->device copy is not allowed while a hipGraph is being captured
                    self._ids_cache[gkey] = torch.tensor(scales, device=dev)
                guidance = self._ids_cache[gkey]
            else:                                         # fresh draws every step: one small H2D copy (not capturable into a hipGraph)
                guidance = torch.tensor(scales, device=dev)
        key = (Hh, Ww, prepared_batch["prompt_embeds"].shape[1])
        if key not in self._ids_cache:
            self._ids_cache[key] = (prepare_latent_image_ids(B, Hh, Ww, dev, BF16),
                                    torch.zeros(prepared_batch["prompt_embeds"].shape[1], 3, device=dev, dtype=torch.float32))
        img_ids, text_ids = self._ids_cache[key]
        rope_key = ("flux_ids",) + key                    # names the id layout for the engine's RoPE-table cache (never tensor identity)
        # "divide it by 1000 for now because we scale it by 1000 in the transformer model" (flux/model.py:739-745, 790)
        ts = prepared_batch["timesteps"].to(device=dev, dtype=torch.float32)
        if ts.ndim == 2:                                  # tokenwise: one timestep per packed image token (flux/model.py:582-595)
            seq = (Hh // 2) * (Ww // 2)
            if ts.shape[1] != seq:
                raise ValueError(f"Flux expected tokenwise timesteps with sequence length {seq}, got {ts.shape[1]}.")
            if ts.shape[0] == 1:
                ts = ts.expand(B, -1)
            elif ts.shape[0] != B:
                raise ValueError(f"Flux expected tokenwise timesteps for batch size {B}, got {ts.shape[0]}.")
        prepared_batch["timesteps"] = ts / 1000.0
        # Kontext (flux/model.py:762-778, 602-618): the packed reference-image tokens are appended to the scene tokens, their position ids to the image ids, and they are
        # conditioned on t = 0 — which makes the timesteps tokenwise: [t ... t | 0 ... 0] per sample
        cond_seq, cond_ids = prepared_batch.get("conditioning_packed_latents"), prepared_batch.get("conditioning_ids")
        use_cond = cond_seq is not None
        scene_len = packed.shape[1]
        if use_cond:
            if cond_ids is None:
                raise ValueError("conditioning_packed_latents needs conditioning_ids (the reference-image tokens' position ids)")
            ts = prepared_batch["timesteps"]
            if ts.ndim == 1:
                ts = ts[:, None].expand(-1, scene_len)
            prepared_batch["timesteps"] = torch.cat([ts, torch.zeros(B, cond_seq.shape[1], device=dev, dtype=torch.float32)], dim=1)
            packed = torch.cat([packed, cond_seq.to(device=dev, dtype=BF16)], dim=1)
            ids_b = img_ids[None].expand(B, -1, -1) if img_ids.dim() == 2 else img_ids
            # the reference-image ids are data: their layout is named by a digest of their contents when they arrive on the host, otherwise the engine
            # computes the tables from the ids on the device every step (no cache entry, no host read)
            rope_key = rope_key + (hashlib.blake2b(cond_ids.detach().float().contiguous().numpy().tobytes(), digest_size=16).hexdigest(),) \
                if cond_ids.device.type == "cpu" else None
            img_ids = torch.cat([ids_b, cond_ids.to(device=dev, dtype=ids_b.dtype)], dim=1)
        attention_mask = None
        if getattr(self.config, "flux_attention_masked_training", False):          # flux/model.py:813-823
            attention_mask = prepared_batch.get("encoder_attention_mask")
            if attention_mask is None:
                raise ValueError("No attention mask was discovered when attempting validation - this means you need to recreate your text embed cache.")
            if attention_mask.dim() == 3 and attention_mask.size(1) == 1:
                attention_mask = attention_mask.squeeze(1)          # [B, 1, S] -> [B, S]
        model_pred = self.model(
            hidden_states=packed,
            timestep=prepared_batch["timesteps"],
            guidance=guidance,
            pooled_projections=prepared_batch["add_text_embeds"].to(device=dev, dtype=BF16),
            encoder_hidden_states=prepared_batch["prompt_embeds"].to(device=dev, dtype=BF16),
            txt_ids=text_ids,
            img_ids=img_ids,
            joint_attention_kwargs=None,
            return_dict=False,
            attention_mask=attention_mask,
            rope_layout_key=rope_key,
        )[0]
        if use_cond and getattr(self.config, "model_flavour", None) == "kontext":      # drop the reference-image tokens before unpacking (flux/model.py:844-847)
            model_pred = model_pred[:, :scene_len, :]
        return {
            "model_prediction": _UnpackFn.apply(model_pred, Hh * 8, Ww * 8),
            "crepa_hidden_states": None,
            "hidden_states_buffer": None,
        }


class _UnpackFn(torch.autograd.Function):
    """unpack_latents with pack_latents as its backward (both are pure permutes, K3)"""

    @staticmethod
    def forward(ctx, packed, height, width):
        return unpack_latents(packed.contiguous(), height=height, width=width, vae_scale_factor=16)

    @staticmethod
    def backward(ctx, g):
        return ops.flux_pack(g.to(BF16).contiguous()), None, None


ModelRegistry.register("flux", Flux)
