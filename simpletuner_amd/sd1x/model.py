"""Stable Diffusion 1.x / 2.x model-family plugins — drop-in for simpletuner/helpers/models/sd1x/model.py on MI355X.

`StableDiffusion1` (epsilon) and `StableDiffusion2` (v-prediction) as the reference declares them (sd1x/model.py:33-60, 351-380) and their step-path
call `(noisy_latents, timesteps, encoder_hidden_states, return_dict=False)[0]` (sd1x/model.py:224-270) — BASELINE.json configs[0] is the SD 1.5
UNet with LoRA rank 16 at 512^2.  The UNet is the same engine as SDXL's with the SD1.5 architecture: 4 levels (320/640/1280/1280), 8 heads per
attention (widths 40 / 80 -> zero-padded heads on the flash kernels, 160 -> the unfused per-head path at <= 256 tokens), conv proj_in/out,
no addition embedding, text width 768.
"""
from __future__ import annotations

import torch

from ..foundation import ModelFoundation, ModelRegistry, ModelTypes, PredictionTypes
from ..unet.unet import UNet2DConditionModel

BF16 = torch.bfloat16

SD15_ARCH = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 transformer_layers_per_block=(1, 1, 1, 1), attention_head_dim=(8, 8, 8, 8), cross_attention_dim=768, use_linear_projection=False,
                 addition_embed_type=None, sample_size=64)


class StableDiffusion1(ModelFoundation):
    NAME = "Stable Diffusion 1.x"
    PREDICTION_TYPE = PredictionTypes.EPSILON
    MODEL_TYPE = ModelTypes.UNET
    MODEL_CLASS = UNet2DConditionModel
    MODEL_SUBFOLDER = "unet"
    LATENT_CHANNEL_COUNT = 4
    TEXT_EMBED_FIELDS = (("prompt_embeds", "prompt_embeds", 3),)                     # sd1x/model.py:112-134
    VAE_CONFIG = dict(latent_channels=4, scaling_factor=0.18215)
    DEFAULT_MODEL_FLAVOUR = "1.5"
    DEFAULT_LORA_TARGET = ["to_k", "to_q", "to_v", "to_out.0"]
    HUGGINGFACE_PATHS = {"1.5": "stable-diffusion-v1-5/stable-diffusion-v1-5", "1.4": "CompVis/stable-diffusion-v1-4"}
    ARCH = SD15_ARCH

    def __init__(self, config, accelerator):
        super().__init__(config, accelerator)
        if getattr(config, "prediction_type", None):                           # sd1x/model.py:325
            self.PREDICTION_TYPE = PredictionTypes.from_str(config.prediction_type)

    def load_model(self, state_dict=None, **arch):
        a = dict(self.ARCH)
        a.update(arch)
        self.model = UNet2DConditionModel(device=self.accelerator.device, **a)
        if state_dict is not None:
            self.model.load_diffusers_state(state_dict)
        else:
            self.model.init_synthetic(seed=int(getattr(self.config, "seed", 42) or 42))
        self.setup_training_noise_schedule()
        return self.model

    def add_lora_adapter(self):
        comp = self.unwrap_model(self.model)
        return comp.add_lora_adapter(rank=int(self.config.lora_rank), alpha=getattr(self.config, "lora_alpha", None),
                                     seed=int(getattr(self.config, "seed", 42) or 42) + 7, init_b_std=float(getattr(self.config, "lora_init_b_std", 0.0)))

    def enable_full_finetune(self):
        return self.unwrap_model(self.model).enable_full_finetune()

    def _convert_lora_state_dict_to_comfyui(self, weights: dict, *, adapter_metadata=None, component_adapter_metadata=None) -> dict:
        """sd1x/model.py:51-65: SD-family ComfyUI files use kohya names (`lora_unet_<module path with _>.lora_down/.lora_up.weight` + `.alpha`)"""
        from ..training.lora_keys import convert_diffusers_to_comfyui_sd_lora
        return convert_diffusers_to_comfyui_sd_lora(weights, adapter_metadata=adapter_metadata, component_adapter_metadata=component_adapter_metadata, sdxl=False)

    def _model_predict_single(self, prepared_batch: dict):
        """sd1x/model.py:224-270"""
        dev = self.accelerator.device
        model_pred = self.model(prepared_batch["noisy_latents"].to(device=dev, dtype=BF16), prepared_batch["timesteps"],
                                prepared_batch["encoder_hidden_states"].to(device=dev, dtype=BF16), return_dict=False)[0]
        return {"model_prediction": model_pred, "hidden_states_buffer": None, "urepa_hidden_states": None}


class StableDiffusion2(StableDiffusion1):
    NAME = "Stable Diffusion 2.x"
    PREDICTION_TYPE = PredictionTypes.V_PREDICTION
    DEFAULT_MODEL_FLAVOUR = "2.1"
    HUGGINGFACE_PATHS = {"2.1": "stabilityai/stable-diffusion-2-1"}
    # SD 2.1: 5/10/20/20 heads of width 64, linear projections, OpenCLIP text width 1024
    ARCH = dict(SD15_ARCH, attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True, sample_size=96)


ModelRegistry.register("sd1x", StableDiffusion1)
ModelRegistry.register("sd2x", StableDiffusion2)
