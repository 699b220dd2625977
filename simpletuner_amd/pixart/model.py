"""PixArt-Sigma model-family plugin — drop-in for simpletuner/helpers/models/pixart/model.py on MI355X.

Step-path methods of the reference plugin (pixart/model.py:44-60, 274-319, 360-379, 399-458): `prepare_batch` (DDPM epsilon objective),
`model_predict` -> the transformer's 8-channel output `.chunk(2, dim=1)[0]` (learned variance dropped), `controlnet_predict` -> the
ControlNet-Transformer wrapper on `conditioning_latents` (scaled by `controlnet_conditioning_scale`), `_build_added_cond_kwargs`
(resolution / aspect_ratio defaults from the latent shape).  BASELINE.json configs[4] trains the ControlNet branch.
"""
from __future__ import annotations

import torch

from ..foundation import ModelFoundation, ModelRegistry, ModelTypes, PredictionTypes
from .transformer import PixArtSigmaControlNetTransformerModel, PixArtTransformer2DModel

BF16 = torch.bfloat16


class PixartSigma(ModelFoundation):
    NAME = "PixArt Sigma"
    PREDICTION_TYPE = PredictionTypes.EPSILON
    MODEL_TYPE = ModelTypes.TRANSFORMER
    MODEL_CLASS = PixArtTransformer2DModel
    MODEL_SUBFOLDER = "transformer"
    LATENT_CHANNEL_COUNT = 4
    TEXT_EMBED_FIELDS = (("prompt_embeds", "prompt_embeds", 3), ("attention_mask", "prompt_attention_mask", 2))      # pixart/model.py:194-224
    COMFYUI_LORA_PRESERVE_COMPONENT_PREFIXES = {"transformer"}      # pixart/model.py:51
    VAE_CONFIG = dict(latent_channels=4, scaling_factor=0.13025)
    DEFAULT_MODEL_FLAVOUR = "900M-1024-v0.6"
    DEFAULT_LORA_TARGET = ["to_k", "to_q", "to_v", "to_out.0"]
    HUGGINGFACE_PATHS = {"900M-1024-v0.6": "terminusresearch/pixart-900m-1024-ft-v0.6", "600M-2048": "PixArt-alpha/PixArt-Sigma-XL-2-2K-MS"}

    def load_model(self, state_dict=None, **arch):
        self.model = PixArtTransformer2DModel(device=self.accelerator.device, **arch)
        if state_dict is not None:
            self.model.load_flat_state(state_dict)
        else:
            self.model.init_synthetic(seed=int(getattr(self.config, "seed", 42) or 42))
        self.controlnet = None
        self.setup_training_noise_schedule()
        return self.model

    def controlnet_init(self, num_layers: int = 13, synthetic_adapter: bool = False):
        """pixart/model.py controlnet_init: adapter blocks copied from the trunk (`from_transformer`), zero before/after projections"""
        self.controlnet = PixArtSigmaControlNetTransformerModel(self.unwrap_model(self.model), num_layers=num_layers, init_from_transformer=True)
        if synthetic_adapter:
            self.controlnet.init_adapter_synthetic(seed=int(getattr(self.config, "seed", 42) or 42) + 11)
        return self.controlnet

    def get_trained_component(self, base_model: bool = False, unwrap_model: bool = True):
        comp = self.controlnet if (self.controlnet is not None and not base_model) else self.model
        return self.unwrap_model(comp) if unwrap_model else comp

    def add_lora_adapter(self):
        """pixart/model.py:59 DEFAULT_LORA_TARGET (to_k, to_q, to_v, to_out.0 of attn1 and attn2 in every trunk block)"""
        if self.controlnet is not None:
            raise NotImplementedError("PixArt: a LoRA on the trunk together with the ControlNet branch is not built on the st355 path")
        comp = self.unwrap_model(self.model)
        return comp.add_lora_adapter(rank=int(self.config.lora_rank), alpha=getattr(self.config, "lora_alpha", None),
                                     seed=int(getattr(self.config, "seed", 42) or 42) + 7, init_b_std=float(getattr(self.config, "lora_init_b_std", 0.0)))

    def _build_added_cond_kwargs(self, prepared_batch: dict) -> dict:
        """pixart/model.py:360-379"""
        dev = self.accelerator.device
        lat = prepared_batch["noisy_latents"]
        B, h, w = lat.shape[0], lat.shape[-2], lat.shape[-1]
        res = prepared_batch.get("resolution")
        ar = prepared_batch.get("aspect_ratio")
        if res is None or ar is None:           # the defaults are cached per latent shape: a host->device copy is not allowed while a hipGraph is being captured
            cache = self.__dict__.setdefault("_cond_cache", {})
            key = (B, h, w, str(dev))
            if key not in cache:
                cache[key] = (torch.tensor([[h, w]], device=dev).expand(B, -1), torch.tensor([[float(h / w)]], device=dev).expand(B, -1))
            res = cache[key][0] if res is None else res.to(device=dev)
            ar = cache[key][1] if ar is None else ar.to(device=dev)
        else:
            res, ar = res.to(device=dev), ar.to(device=dev)
        return {"resolution": res, "aspect_ratio": ar}

    def model_predict(self, prepared_batch: dict):
        if self.controlnet is not None and prepared_batch.get("conditioning_latents") is not None:
            return self.controlnet_predict(prepared_batch)
        return super().model_predict(prepared_batch)

    def _model_predict_single(self, prepared_batch: dict):
        self._require_per_sample_timesteps(prepared_batch, tokenwise_ok=True)      # [B] or tokenwise [B, S] handed through unchanged (tests/test_pixart_model.py:91-115)
        dev = self.accelerator.device
        if prepared_batch["noisy_latents"].shape[1] != self.LATENT_CHANNEL_COUNT:
            raise ValueError(f"{self.NAME} requires a latent size of {self.LATENT_CHANNEL_COUNT} channels. Ensure you are using the correct VAE cache path.")
        out = self.model(prepared_batch["noisy_latents"].to(device=dev, dtype=BF16),
                         encoder_hidden_states=prepared_batch["encoder_hidden_states"].to(device=dev, dtype=BF16),
                         timestep=prepared_batch["timesteps"], encoder_attention_mask=prepared_batch["encoder_attention_mask"].to(device=dev, dtype=BF16),
                         added_cond_kwargs=self._build_added_cond_kwargs(prepared_batch), return_dict=False)[0].chunk(2, dim=1)[0]
        return {"model_prediction": out, "crepa_hidden_states": None, "hidden_states_buffer": None}

    def _controlnet_predict_single(self, prepared_batch: dict) -> dict:
        """pixart/model.py:399-458"""
        self._require_per_sample_timesteps(prepared_batch)
        dev = self.accelerator.device
        cond = prepared_batch.get("conditioning_latents")
        if cond is None:
            raise ValueError("conditioning_latents must be provided for ControlNet training")
        cond = cond.to(device=dev, dtype=BF16)
        if cond.shape[1] != self.LATENT_CHANNEL_COUNT:
            raise ValueError(f"ControlNet conditioning latents must have {self.LATENT_CHANNEL_COUNT} channels. Got {cond.shape[1]} channels.")
        scale = getattr(self.config, "controlnet_conditioning_scale", 1.0)
        if scale != 1.0:
            cond = cond * scale
        out = self.controlnet(prepared_batch["noisy_latents"].to(device=dev, dtype=BF16),
                              encoder_hidden_states=prepared_batch["encoder_hidden_states"].to(device=dev, dtype=BF16),
                              timestep=prepared_batch["timesteps"], encoder_attention_mask=prepared_batch["encoder_attention_mask"].to(device=dev, dtype=BF16),
                              controlnet_cond=cond, added_cond_kwargs=self._build_added_cond_kwargs(prepared_batch), return_dict=False)[0]
        if out.shape[1] == self.LATENT_CHANNEL_COUNT * 2:
            out = out.chunk(2, dim=1)[0]
        return {"model_prediction": out}


ModelRegistry.register("pixart_sigma", PixartSigma)
