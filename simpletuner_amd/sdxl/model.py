"""SDXL model-family plugin — drop-in for simpletuner/helpers/models/sdxl/model.py on MI355X.

Same class attributes and step-path methods as the reference plugin (sdxl/model.py:40-120, 306-373): `SDXL(config, accelerator)`,
`prepare_batch` (DDPM epsilon objective: discrete timesteps + `noise_schedule.add_noise`, common.py:5983-6002), `model_predict ->
{"model_prediction": [B,4,H,W]}` calling the UNet positionally `(noisy_latents, timesteps, encoder_hidden_states, add_text_embeds,
added_cond_kwargs={"text_embeds","time_ids"}, return_dict=False)[0]`, `loss_with_logs`, `get_trained_component`.  Full fine-tune
(BASELINE.json configs[1]) and LoRA on the attention projections (the metric's "SDXL-LoRA").
"""
from __future__ import annotations

import torch

from ..foundation import ModelFoundation, ModelRegistry, ModelTypes, PredictionTypes
from ..unet.unet import UNet2DConditionModel

BF16 = torch.bfloat16


class SDXL(ModelFoundation):
    NAME = "Stable Diffusion XL"
    PREDICTION_TYPE = PredictionTypes.EPSILON
    MODEL_TYPE = ModelTypes.UNET
    MODEL_CLASS = UNet2DConditionModel
    MODEL_SUBFOLDER = "unet"
    LATENT_CHANNEL_COUNT = 4
    VAE_CONFIG = dict(latent_channels=4, scaling_factor=0.13025)
    DEFAULT_MODEL_FLAVOUR = "base-1.0"
    DEFAULT_LORA_TARGET = ["to_k", "to_q", "to_v", "to_out.0"]
    HUGGINGFACE_PATHS = {"base-1.0": "stabilityai/stable-diffusion-xl-base-1.0"}

    def load_model(self, state_dict=None, **arch):
        self.model = UNet2DConditionModel(device=self.accelerator.device, **arch)
        if state_dict is not None:
            self.model.load_diffusers_state(state_dict)
        else:
            self.model.init_synthetic(seed=int(getattr(self.config, "seed", 42) or 42))
        self.setup_training_noise_schedule()
        return self.model

    def add_lora_adapter(self):
        """peft LoRA on DEFAULT_LORA_TARGET (to_k,to_q,to_v,to_out.0 of every attn1 / attn2): the "SDXL-LoRA" workload of BASELINE.json's metric"""
        comp = self.unwrap_model(self.model)
        return comp.add_lora_adapter(rank=int(self.config.lora_rank), alpha=getattr(self.config, "lora_alpha", None),
                                     seed=int(getattr(self.config, "seed", 42) or 42) + 7, init_b_std=float(getattr(self.config, "lora_init_b_std", 0.0)))

    def enable_full_finetune(self):
        return self.unwrap_model(self.model).enable_full_finetune()

    def _convert_lora_state_dict_to_comfyui(self, weights: dict, *, adapter_metadata=None, component_adapter_metadata=None) -> dict:
        """sdxl/model.py:61-75: SD-family ComfyUI files use kohya names (`lora_unet_<module path with _>.lora_down/.lora_up.weight` + `.alpha`)"""
        from ..training.lora_keys import convert_diffusers_to_comfyui_sd_lora
        return convert_diffusers_to_comfyui_sd_lora(weights, adapter_metadata=adapter_metadata, component_adapter_metadata=component_adapter_metadata, sdxl=True)

    def _model_predict_single(self, prepared_batch: dict):
        """sdxl/model.py:306-373"""
        dev = self.accelerator.device
        model_pred = self.model(
            prepared_batch["noisy_latents"].to(device=dev, dtype=BF16),
            prepared_batch["timesteps"],
            prepared_batch["encoder_hidden_states"].to(device=dev, dtype=BF16),
            prepared_batch["add_text_embeds"].to(device=dev, dtype=BF16),
            added_cond_kwargs=prepared_batch["added_cond_kwargs"],
            return_dict=False,
        )[0]
        return {"model_prediction": model_pred, "hidden_states_buffer": None, "urepa_hidden_states": None}


ModelRegistry.register("sdxl", SDXL)
