"""SD3 model-family plugin — drop-in for simpletuner/helpers/models/sd3/model.py on MI355X.

Same class attributes and step-path methods as the reference plugin (sd3/model.py:111-125, 540-570): `SD3(config, accelerator)`,
`prepare_batch` (flow matching, common.py:5862-6041), `model_predict -> {"model_prediction": [B,16,H,W]}` with the timesteps passed
to the transformer in 0..1000 (sd3/model.py:542 — no /1000, unlike Flux), `loss_with_logs`, `get_trained_component`,
`add_lora_adapter` (DEFAULT_LORA_TARGET to_k,to_q,to_v,to_out.0).
"""
from __future__ import annotations

import torch

from ..foundation import ModelFoundation, ModelRegistry, ModelTypes, PredictionTypes
from .transformer import SD3Transformer2DModel

BF16 = torch.bfloat16


class SD3(ModelFoundation):
    NAME = "Stable Diffusion 3.x"
    PREDICTION_TYPE = PredictionTypes.FLOW_MATCHING
    MODEL_TYPE = ModelTypes.TRANSFORMER
    MODEL_CLASS = SD3Transformer2DModel
    MODEL_SUBFOLDER = "transformer"
    LATENT_CHANNEL_COUNT = 16
    COMFYUI_LORA_PRESERVE_COMPONENT_PREFIXES = {"transformer"}      # sd3/model.py:117
    VAE_CONFIG = dict(latent_channels=16, scaling_factor=1.5305, shift_factor=0.0609, use_quant_conv=False)
    DEFAULT_MODEL_FLAVOUR = "medium"
    DEFAULT_LORA_TARGET = ["to_k", "to_q", "to_v", "to_out.0"]
    HUGGINGFACE_PATHS = {"medium": "stabilityai/stable-diffusion-3-medium-diffusers", "large": "stabilityai/stable-diffusion-3.5-large"}

    def load_model(self, state_dict=None, **arch):
        self.model = SD3Transformer2DModel(device=self.accelerator.device, **arch)
        if state_dict is not None:
            self.model.load_flat_state(state_dict)
        else:
            self.model.init_synthetic(seed=int(getattr(self.config, "seed", 42) or 42))
        return self.model

    def add_lora_adapter(self):
        if getattr(self.config, "model_type", "lora") != "lora":
            raise RuntimeError("model_type == 'full' trains every transformer parameter: call enable_full_finetune() (or freeze_components()) instead of add_lora_adapter()")
        comp = self.unwrap_model(self.model)
        params = comp.add_lora_adapter(rank=int(self.config.lora_rank), alpha=getattr(self.config, "lora_alpha", None),
                                       targets="default", seed=int(getattr(self.config, "seed", 42) or 42) + 7,
                                       init_b_std=float(getattr(self.config, "lora_init_b_std", 0.0)))
        comp.prepare_for_training()
        return params

    def enable_full_finetune(self):
        """model_type == "full" (BASELINE.json configs[3]): every transformer parameter trains (bf16 params + bf16 grads in two arenas)"""
        return self.unwrap_model(self.model).enable_full_finetune()

    def _model_predict_single(self, prepared_batch: dict):
        """sd3/model.py:540-570"""
        self._require_per_sample_timesteps(prepared_batch, tokenwise_ok=True)      # [B] or tokenwise [B, S_img], handed through unchanged (sd3/model.py:542)
        dev = self.accelerator.device
        model_pred = self.model(
            hidden_states=prepared_batch["noisy_latents"].to(device=dev, dtype=BF16),
            timestep=prepared_batch["timesteps"].to(device=dev, dtype=torch.float32),
            encoder_hidden_states=prepared_batch["encoder_hidden_states"].to(device=dev, dtype=BF16),
            pooled_projections=prepared_batch["add_text_embeds"].to(device=dev, dtype=BF16),
            return_dict=False,
        )[0]
        return {"model_prediction": model_pred, "crepa_hidden_states": None, "hidden_states_buffer": None}


ModelRegistry.register("sd3", SD3)
