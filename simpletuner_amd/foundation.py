"""Model-family plugin surface, mirroring the reference's duck-typed contract:

    ModelRegistry.register / get / model_families        simpletuner/helpers/models/registry.py:54-98
    PredictionTypes / ModelTypes                         simpletuner/helpers/models/common.py:368-392
    ModelFoundation.prepare_batch                        simpletuner/helpers/models/common.py:5862-6041
                   .sample_flow_sigmas                   simpletuner/helpers/models/common.py:4994-5090
                   ._prepare_flow_noisy_latents          simpletuner/helpers/models/common.py:4975-4992
                   .get_prediction_target / flow target  simpletuner/helpers/models/common.py:4610-4658
                   .loss / loss_with_logs / aux loss     simpletuner/helpers/models/common.py:6217-6430, xm_mixin.py:476-485
    apply_flow_schedule_shift                            simpletuner/helpers/training/custom_schedule.py:443-478

Same method names, argument meaning, batch-dict keys and error behaviour; the bulk arithmetic (noising, target, MSE and
its gradient) runs in libst355 (ops.flow_noise_mix / ops.mse_loss).  Per-sample scalars ([B]-sized sigma math) stay in torch.
Only the options used by the BASELINE configs are implemented; anything else raises NotImplementedError (never a silent
different path).
"""
from __future__ import annotations

import math
from enum import Enum
from typing import Any, Dict, Optional, Type

import torch

from . import ops
from .xm import ExplorativeModelingConfig, ExplorativeModelingMixin

BF16 = torch.bfloat16


class PredictionTypes(Enum):
    EPSILON = "epsilon"
    SAMPLE = "sample"
    V_PREDICTION = "v_prediction"
    FLOW_MATCHING = "flow_matching"

    @staticmethod
    def from_str(label):
        if label in ("eps", "epsilon"):
            return PredictionTypes.EPSILON
        if label in ("vpred", "v_prediction", "v-prediction"):
            return PredictionTypes.V_PREDICTION
        if label in ("sample", "x_prediction", "x-prediction"):
            return PredictionTypes.SAMPLE
        if label in ("flow", "flow_matching", "flow-matching"):
            return PredictionTypes.FLOW_MATCHING
        raise NotImplementedError


class ModelTypes(Enum):
    UNET = "unet"
    TRANSFORMER = "transformer"
    VAE = "vae"
    TEXT_ENCODER = "text_encoder"


class ModelRegistry:
    """registry.py:54-98 (the metadata-JSON lazy path is the reference's own; a drop-in only needs register/get)."""
    _registry: Dict[str, Type[Any]] = {}

    @classmethod
    def register(cls, family: str, model_class: Type[Any]) -> None:
        cls._registry[family.lower()] = model_class

    @classmethod
    def get(cls, family: str):
        return cls._registry.get(family.lower())

    @classmethod
    def model_families(cls) -> Dict[str, Type[Any]]:
        return dict(cls._registry)


def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096, base_shift: float = 0.5, max_shift: float = 1.15):
    """diffusers.pipelines.flux.pipeline_flux.calculate_shift (imported at flux/__init__.py:5)"""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def apply_flow_schedule_shift(args, noise_scheduler, sigmas, noise):
    """custom_schedule.py:443-478"""
    shift = None
    if getattr(args, "flow_schedule_shift", None) is not None and args.flow_schedule_shift > 0:
        shift = args.flow_schedule_shift
    elif getattr(args, "flow_schedule_auto_shift", False):
        if noise.ndim == 5:
            num_frames, height, width = noise.shape[-3:]
        else:
            num_frames = 1
            height, width = noise.shape[-2:]
        cfg = getattr(noise_scheduler, "config", None)
        patch_size = getattr(cfg, "patch_size", 2) or 2
        if patch_size <= 0:
            patch_size = 2
        seq_len = num_frames * (height // patch_size) * (width // patch_size)
        mu = calculate_shift(seq_len, getattr(cfg, "base_image_seq_len", 256), getattr(cfg, "max_image_seq_len", 4096),
                             getattr(cfg, "base_shift", 0.5), getattr(cfg, "max_shift", 1.15))
        shift = math.exp(mu)
    if shift is not None:
        sigmas = (sigmas * shift) / (1 + (shift - 1) * sigmas)
    return sigmas


def enforce_zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    """`rescale_betas_zero_snr` (custom_schedule.py:157-175; "Common Diffusion Noise Schedules and Sample Steps are Flawed", alg. 1): shift
    sqrt(alphas_cumprod) so the last step carries no signal, rescale so the first step keeps its value, convert back to betas"""
    root = (1 - betas).cumprod(0).sqrt()
    first, last = root[0].clone(), root[-1].clone()
    root = (root - last) * (first / (first - last))
    bar = root ** 2
    return 1 - torch.cat([bar[0:1], bar[1:] / bar[:-1]])


class DDPMSchedule:
    """the training-side arithmetic of diffusers' DDPMScheduler as the reference configures it for SD1.5 / SDXL (common.py:4529-4550; the
    checkpoint's scheduler_config.json: beta_schedule "scaled_linear", beta_start 0.00085, beta_end 0.012, 1000 steps)."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012, beta_schedule: str = "scaled_linear",
                 prediction_type: str = "epsilon", device=None, rescale_betas_zero_snr: bool = False):
        from types import SimpleNamespace
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            betas = enforce_zero_terminal_snr(betas)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type, beta_schedule=beta_schedule,
                                      rescale_betas_zero_snr=bool(rescale_betas_zero_snr))
        self._acp_dev = {}
        self._sa = self.alphas_cumprod.sqrt().to(device)
        self._sb = (1.0 - self.alphas_cumprod).sqrt().to(device)

    def mix_coefficients(self, timesteps):
        return self._sa[timesteps].contiguous(), self._sb[timesteps].contiguous()

    def snr(self, timesteps):
        """compute_snr (min_snr_gamma.py:4-41): (alpha/sigma)^2 = acp / (1 - acp)"""
        a = self.acp_on(timesteps.device)[timesteps.long()].float()
        return a / (1.0 - a)

    def acp_on(self, device):
        """alphas_cumprod resident on `device` (copied once): the SNR / huber paths run inside hipGraph capture, where an H2D copy is illegal"""
        key = str(torch.device(device))
        hit = self._acp_dev.get(key)
        if hit is None:
            hit = self._acp_dev[key] = self.alphas_cumprod.to(device)
        return hit

    def min_snr_weights(self, timesteps, gamma: float, v_prediction: bool):
        """common.py:6363-6397: min(snr, gamma) / snr  (epsilon)  or  / (snr + 1)  (v-prediction), one weight per sample"""
        snr = self.snr(timesteps)
        return torch.minimum(snr, torch.full_like(snr, float(gamma))) / (snr + 1.0 if v_prediction else snr)

    @staticmethod
    def timestep_weights(config, T: int) -> torch.Tensor:
        """generate_timestep_weights (custom_schedule.py:61-100): uniform, or a multiplier on the later / earlier / [begin, end) timesteps"""
        w = torch.ones(T)
        strat = getattr(config, "timestep_bias_strategy", "none") if config is not None else "none"
        n = int(float(getattr(config, "timestep_bias_portion", 0.25)) * T) if config is not None else 0
        if strat == "later":
            idx = slice(-n, None)
        elif strat == "earlier":
            idx = slice(0, n)
        elif strat == "range":
            lo, hi = int(config.timestep_bias_begin), int(config.timestep_bias_end)
            if lo < 0 or hi > T:
                raise ValueError("timestep_bias_begin / timestep_bias_end must lie inside [0, num_train_timesteps]")
            idx = slice(lo, hi)
        else:
            return w
        mult = float(getattr(config, "timestep_bias_multiplier", 1.0))
        if mult <= 0:
            raise ValueError("timestep_bias_multiplier must be positive (use timestep_bias_strategy=none to disable the bias)")
        w[idx] *= mult
        return w / w.sum()

    def sample_timesteps(self, bsz: int, segmented: bool = True, weights: Optional[torch.Tensor] = None, refiner_training: bool = False,
                         refiner_invert_schedule: bool = False, refiner_strength: float = 0.2):
        """weights: generate_timestep_weights (uniform by default); bsz > 1: one draw from each of bsz equal segments, high to low
        (segmented_timestep_selection, custom_schedule.py:18-58, same draw order as the reference); SDXL-refiner training restricts the range to
        the low-noise tail [0, strength*T) — or, inverted, to [strength*T, T) (:21-31).  Drawn on the host: no device sync."""
        T = self.config.num_train_timesteps
        weights = torch.ones(T) if weights is None else weights.clone()
        if bsz == 1 or not segmented:
            return torch.multinomial(weights, bsz, replacement=True).long()
        hi, lo = T - 1, 0
        if refiner_training:
            hi, lo = (T - 1, int(refiner_strength * T)) if refiner_invert_schedule else (int(T * refiner_strength) - 1, 0)
        seg = max((hi - lo + 1) // bsz, 1)
        out = []
        for i in range(bsz):
            start = hi - i * seg
            end = max(start - seg, lo) if i != bsz - 1 else lo
            w = weights[end:start + 1]
            w /= w.sum()                                   # in place on the slice, as the reference does (:50)
            out.append(end + int(torch.multinomial(w, 1).item()))
        return torch.tensor(out, dtype=torch.long)


DATA_BACKEND_CONFIGS = {}


def get_data_backend_config(backend_id) -> dict:
    """stand-in for StateTracker.get_data_backend_config (state_tracker.py): unknown / missing ids have an empty config"""
    src = DATA_BACKEND_CONFIGS
    cfg = src(backend_id) if callable(src) else src.get(backend_id)
    return cfg or {}


def _process_rank(accelerator=None) -> int:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    return int(getattr(accelerator, "process_index", 0) or 0)


class ModelFoundation(ExplorativeModelingMixin):
    """the subset of common.py's ModelFoundation that the step loop touches (trainer.py:6951-7568)."""
    NAME = "foundation"
    PREDICTION_TYPE = PredictionTypes.FLOW_MATCHING
    MODEL_TYPE = ModelTypes.TRANSFORMER
    MODEL_CLASS = None
    MODEL_SUBFOLDER = "transformer"
    LATENT_CHANNEL_COUNT = 16
    DEFAULT_LORA_TARGET = ["to_k", "to_q", "to_v", "to_out.0"]
    DDP_FIND_UNUSED_PARAMETERS = False
    PIPELINE_CLASSES: Dict[str, Any] = {}        # validation pipelines are outside this tier (sampling.py holds the denoising loop)
    TEXT_ENCODER_CONFIGURATION: Dict[str, Any] = {}   # text encoders run offline into the embed cache (training/cache_io.py reads it)
    AUTOENCODER_CLASS = None                     # set per family below (lazy: the class needs the device library)
    VAE_CONFIG: Dict[str, Any] = {}              # constructor arguments of the family's AutoencoderKL (upstream `vae/config.json` values)

    def __init__(self, config, accelerator):
        self.config = config
        self.accelerator = accelerator
        self.model = None
        # components the reference Trainer reads / clears on the plugin (trainer.py:2592-2593, 2676-2677, 2914-2917, 3050, 4375): text encoders run
        # OFFLINE into the embed cache on this path (training/cache_io.py), so the lists stay empty and the unload hooks have nothing to move
        self.vae = None
        self.text_encoders: list = []
        self.tokenizers: list = []
        self.controlnet = None
        self.pipelines: dict = {}
        self.pipeline = None
        self.ema_model = None
        self.noise_schedule = None
        self._noise_step = 0            # noising passes issued (diagnostic)
        self._noise_offset = 0          # CUMULATIVE Philox counter position: every pass consumes its own, non-overlapping counter range whatever its shape
        self.xm_config = ExplorativeModelingConfig.from_config(config)     # common.py:578

    # ---- latent encode seam (common.py:2653-2772, foundation_mixins.py:66-79) ----
    @classmethod
    def autoencoder_class(cls):
        if cls.AUTOENCODER_CLASS is None:
            from .vae.autoencoder_kl import AutoencoderKL
            return AutoencoderKL
        return cls.AUTOENCODER_CLASS

    def load_vae(self, state_dict=None, move_to_device: bool = True):
        """common.py:2663: build the family's AutoencoderKL (encoder half; diffusers state-dict keys).  Without a state dict the weights are
        synthetic (benchmarks, tests) — there is no hub access here."""
        vae = self.autoencoder_class()(device=self.accelerator.device, **self.VAE_CONFIG)
        vae.load_state_dict(state_dict if state_dict is not None else vae.synthetic_state_dict(int(getattr(self.config, "seed", 42) or 42), decoder=True))
        self.vae = vae
        return vae

    def get_vae(self):
        if getattr(self, "vae", None) is None:
            self.load_vae()
        return self.vae

    @torch.no_grad()
    def encode_with_vae(self, vae, samples):
        return vae.encode(samples)

    def scale_vae_latents_for_cache(self, latents, vae):
        """foundation_mixins.py:66-79: (z - shift) * scale when the VAE has a shift factor, z * scale otherwise"""
        if vae is None or not hasattr(vae, "config") or latents is None:
            return latents
        shift = getattr(vae.config, "shift_factor", None)
        scale = getattr(self, "AUTOENCODER_SCALING_FACTOR", getattr(vae.config, "scaling_factor", 1.0))
        if shift is not None:
            return (latents - shift) * scale
        if isinstance(latents, torch.Tensor) and hasattr(vae.config, "scaling_factor"):
            return latents * scale
        return latents

    # ---- prediction entry points; XM (off by default) expands the batch with K noise candidates first (flux/model.py:630-636, 940-946) ----
    def model_predict(self, prepared_batch: dict):
        return self._xm_wrapped(self._model_predict_single, prepared_batch)

    def controlnet_predict(self, prepared_batch: dict):
        return self._xm_wrapped(self._controlnet_predict_single, prepared_batch)

    def _xm_wrapped(self, predict, prepared_batch: dict):
        if self._xm_noise_candidates_enabled(prepared_batch):
            self._prepare_xm_noise_candidates(prepared_batch)
            out = predict(prepared_batch)
            out["xm_candidate_count"] = self.xm_config.candidate_count
            return out
        return predict(prepared_batch)

    def _model_predict_single(self, prepared_batch: dict):
        raise NotImplementedError

    def _controlnet_predict_single(self, prepared_batch: dict):
        raise NotImplementedError(f"{self.NAME} has no ControlNet path")

    # ---- component plumbing (common.py:3691, 3781) ----
    def get_trained_component(self, base_model: bool = False, unwrap_model: bool = True):
        return self.unwrap_model(self.model) if unwrap_model else self.model

    def set_prepared_model(self, model, base_model: bool = False):
        """common.py:3691 / trainer.py:4577: the trainer hands back what `accelerator.prepare` returned.  A torch DistributedDataParallel around an st355
        component gets the st355 communication hook (the component's own in-backward exchange, training.ddp_seam) unless it already has one; an
        `St355DistributedDataParallel` wrapper, or the bare component, is stored as is."""
        self.model = model
        if type(model).__name__ == "DistributedDataParallel" and hasattr(model, "register_comm_hook") and getattr(model, "_st355_seam", None) is None:
            from .training.ddp_seam import install_ddp_comm_hook
            install_ddp_comm_hook(model)

    @staticmethod
    def unwrap_model(model):
        """common.py:3700 (accelerator.unwrap_model): strip DDP-style wrappers (`.module`), however nested"""
        seen = 0
        while hasattr(model, "module") and isinstance(getattr(model, "module"), torch.nn.Module) and seen < 4:
            model, seen = model.module, seen + 1
        return model

    def _require_per_sample_timesteps(self, prepared_batch: dict, tokenwise_ok: bool = False, conditioning_ok: bool = False):
        """The reference's DiT plugins also accept TOKENWISE timesteps [B, S] (CREPA self-flow; tests/test_flux_model.py:213-241,
        tests/test_sd3_model.py:179-204, tests/test_pixart_model.py:91-115) and clean conditioning tokens appended at t=0 (Flux Kontext,
        tests/test_flux_model.py:243-272).  Both need per-token modulation rows.  Tokenwise timesteps are built for SD3, Flux and the PixArt trunk (`tokenwise_ok`: their engines run the AdaLN /
        gated-residual kernels with one modulation row per token); the PixArt ControlNet wrapper, and the reference-image tokens everywhere, refuse loudly instead of training on the wrong conditioning."""
        t = prepared_batch["timesteps"]
        if getattr(t, "ndim", 1) == 2 and tokenwise_ok:
            pass
        elif getattr(t, "ndim", 1) != 1:
            raise NotImplementedError(f"tokenwise timesteps {tuple(t.shape)} are not implemented on the st355 path (per-sample [B] only)")
        if prepared_batch.get("conditioning_packed_latents") is not None and not conditioning_ok:
            raise NotImplementedError("conditioning_packed_latents (reference-image tokens) are not implemented on the st355 path")

    # ---- lifecycle hooks the reference Trainer calls on `self.model` outside the step loop (trainer.py:329-330, 2618-2621, 2944, 3243-3292, 3353,
    #      4318, 4376, 4450, 4651, 5539, 6042, 7631-7642, 7749-7754).  tests/test_integration_contract_cpu.py AST-scans that file and fails on any
    #      `self.model.<name>` this class lacks.  Hooks whose feature lives outside the hot path are explicit no-ops or loud refusals. ----
    GLIGEN_LYCORIS_TARGET: list = []

    def check_user_config(self):
        """common.py:2137 (family overrides adjust config in place).  The st355 path checks what it cannot honour: bf16 compute only."""
        wd = getattr(self.config, "weight_dtype", BF16)
        if wd not in (BF16, "bf16", None):
            raise ValueError(f"{self.NAME} (st355): weight_dtype {wd} — the MI355X path computes in bf16 (mixed_precision=bf16)")

    def _mixflow_enabled(self) -> bool:
        return getattr(self.config, "mixflow_enabled", False) is True

    def validate_mixflow_config(self) -> None:
        """common.py:4923-4950: mixflow needs a flow-matching family and excludes the alternative sigma samplers"""
        if not self._mixflow_enabled():
            return
        if self.PREDICTION_TYPE is not PredictionTypes.FLOW_MATCHING:
            raise ValueError("mixflow_enabled requires a flow-matching model family.")
        self._mixflow_gamma()
        conflicts = {"flux_fast_schedule": bool(getattr(self.config, "flux_fast_schedule", False)),
                     "flow_use_uniform_schedule": bool(getattr(self.config, "flow_use_uniform_schedule", False)),
                     "flow_use_beta_schedule": bool(getattr(self.config, "flow_use_beta_schedule", False)),
                     "flow_custom_timesteps": getattr(self.config, "flow_custom_timesteps", None) not in (None, "", "None")}
        active = [k for k, v in conflicts.items() if v]
        if active:
            raise ValueError(f"mixflow_enabled cannot be combined with {', '.join(active)}.")

    def load_text_encoder(self, move_to_device: bool = True):
        """common.py:2949.  Text encoders are not part of the per-step path (embeddings come from the cache, caching/text_embeds.py): nothing is loaded"""
        self.text_encoders, self.tokenizers = [], []

    # ---- text-embed cache record -> validation-pipeline kwargs (the three members besides model_predict that the reference's ABC declares
    # abstract, common.py:1744-1766; a class registered into ModelRegistry must define them to be instantiable) -------------------------------------
    # (cache-record key, pipeline kwarg, rank WITH the batch dimension); per family as written in <family>/model.py `convert_text_embed_for_pipeline`
    TEXT_EMBED_FIELDS = (("prompt_embeds", "prompt_embeds", 3), ("pooled_prompt_embeds", "pooled_prompt_embeds", 2))

    def _convert_text_embed(self, text_embedding: dict, negative: bool) -> dict:
        out = {}
        for src, dst, rank in self.TEXT_EMBED_FIELDS:
            t = text_embedding[src]
            if t.dim() == rank - 1:                  # "Add batch dimension if missing" (e.g. sd3/model.py:392-396)
                t = t.unsqueeze(0)
            out[("negative_" + dst) if negative else dst] = t
        return out

    def convert_text_embed_for_pipeline(self, text_embedding: dict) -> dict:
        """e.g. sd3/model.py:387-401, sdxl/model.py:135-149, sd1x/model.py:112-122, pixart/model.py:194-208 (Flux overrides: prompt_mask)"""
        return self._convert_text_embed(text_embedding, negative=False)

    def convert_negative_text_embed_for_pipeline(self, text_embedding: dict) -> dict:
        """e.g. sd3/model.py:403-417: the same record under the `negative_*` kwarg names (the CFG half of the validation pair)"""
        return self._convert_text_embed(text_embedding, negative=True)

    def _encode_prompts(self, prompts: list, is_negative_prompt: bool = False):
        """common.py:1744-1750.  Running the CLIP / T5 text encoders is outside the per-step path (text-embed caching is offline
        preprocessing): the st355 plugins consume cached embeddings.  Under `integration.register()` with an importable SimpleTuner the registered
        class also derives from the reference's own family class, whose `_encode_prompts` is then the one that runs."""
        for base in type(self).__mro__:
            if base.__module__.startswith("simpletuner.") and "_encode_prompts" in base.__dict__ and not getattr(base.__dict__["_encode_prompts"], "__isabstractmethod__", False):
                return base.__dict__["_encode_prompts"](self, prompts, is_negative_prompt)
        raise NotImplementedError("text encoders are not part of the st355 per-step path: feed cached text embeddings (collate_fn / DirectCacheFeeder)")

    def get_text_encoder(self, index: int):
        """common.py:3130"""
        if self.text_encoders is not None:
            return self.text_encoders[index] if 0 <= index < len(self.text_encoders) else None
        return None

    def unload_text_encoder(self):
        """common.py:3134"""
        self.text_encoders, self.tokenizers = [], []

    def unload_vae(self):
        """common.py:2794"""
        self.vae = None

    def unload(self):
        """common.py:3214: drop every component (device memory returns to the allocator when the last reference dies)"""
        self.unload_vae()
        self.unload_text_encoder()
        self.model = None
        self.controlnet = None
        self.pipelines, self.pipeline = {}, None

    def freeze_components(self):
        """common.py:3700-3709.  Frozen-ness is structural here: base weights are `requires_grad=False` parameters by construction (LoRA) and the
        full fine-tune exposes exactly its arena views; the VAE / ControlNet trunk never carry gradients"""
        if "lora" in str(getattr(self.config, "model_type", "lora")) and self.model is not None:
            for n, p_ in self.unwrap_model(self.model).named_parameters():
                if ".lora_" not in n:
                    p_.requires_grad_(False)
        elif str(getattr(self.config, "model_type", "lora")) == "full" and self.model is not None:
            # the reference leaves a full-rank model as diffusers loaded it — every parameter trainable — and `Trainer._get_trainable_parameters` collects
            # `requires_grad` parameters (trainer.py:3668-3673).  Here trainability is a mode of the component (gradient arena + full backward): entered at the same
            # point of the lifecycle, so an unmodified Trainer needs no extra call
            comp = self.unwrap_model(self.model)
            if hasattr(comp, "enable_full_finetune") and not getattr(comp, "full", False):
                comp.enable_full_finetune()

    def pre_ema_creation(self):
        """common.py:2125: the reference fuses qkv here so EMA shapes line up; fused storage is the only layout on this path"""
        self.fuse_qkv_projections()

    def post_ema_creation(self):
        """common.py:2131"""
        return None

    def post_model_load_setup(self):
        """common.py:3638 / 6704 (ImageModelFoundation).  The reference attaches its representation-alignment regularisers here (LayerSync, internal guidance,
        NextLat, CREPA / U-REPA: hooks into a diffusers module's blocks).  None of them is built on the st355 path: asking for one is refused, otherwise there is
        nothing to set up — defined here so that a reference flow calling it never reaches the reference's regulariser initialisers through the MRO."""
        for flag in ("crepa_enabled", "irepa_enabled", "urepa_enabled", "layersync_enabled", "internal_guidance_enabled", "nextlat_enabled"):
            if getattr(self.config, flag, False):
                raise NotImplementedError(f"{flag}: representation-alignment regularisers hook diffusers modules and are not built on the st355 path")
        return None

    def post_quantization_setup(self):
        """common.py:3650"""
        return None

    def before_accelerator_prepare(self):
        """common.py:3697"""
        return None

    def refresh_representation_alignment_projectors(self):
        """common.py:939: CREPA / U-REPA projectors are not built on this path (no regulariser is attached): nothing to refresh"""
        return None

    def supports_grounding(self) -> bool:
        """common.py:1452.  GLIGEN grounding layers are out of the hot path: never advertised"""
        return False

    def enable_muon_clip_logging(self):
        raise NotImplementedError("optimizer 'muon' is not built on the st355 path")

    def configure_assistant_lora_for_training(self):
        if getattr(self.config, "assistant_lora_path", None) not in (None, "", "None"):
            raise NotImplementedError("assistant LoRA (schnell) is not implemented on the st355 path")

    def configure_group_offload(self):
        raise NotImplementedError("group offload targets small-memory cards; the st355 path keeps everything resident in the 288 GB of HBM")

    def _ramtorch_base_deferred_until_after_quantization(self) -> bool:
        return False

    def apply_ramtorch_to_transformer(self) -> bool:
        return False

    def apply_ramtorch_to_controlnet(self) -> bool:
        return False

    def controlnet_init(self, *a, **k):
        raise NotImplementedError(f"{self.NAME} has no ControlNet path on st355 (PixArt-Sigma has: pixart/model.py)")

    def tread_init(self):
        """<family>/model.py `tread_init` (e.g. sd3/model.py:326-347): hand the trained component a TREADRouter seeded from the run seed and the routes of
        `config.tread_config`.  Built for the components that expose `set_router` (SD3, Flux, the PixArt trunk under LoRA; training/tread.py); the others refuse — never a silent no-op."""
        tc = getattr(self.config, "tread_config", None)
        if not tc or tc.get("routes", None) is None:
            raise ValueError("TREAD training requires you to configure the routes in the TREAD config")
        comp = self.get_trained_component()
        if comp is None or not hasattr(comp, "set_router"):
            raise NotImplementedError(f"tread_init: TREAD routing is not implemented for {self.NAME} on the st355 path (built: SD3, Flux, the PixArt trunk)")
        from .training.tread import TREADRouter
        comp.set_router(TREADRouter(seed=getattr(self.config, "seed", None) or 42, device=self.accelerator.device), tc["routes"])

    def diffusion_blocks_init(self) -> None:
        if getattr(self.config, "diffusion_blocks_config", None):
            raise NotImplementedError("diffusion_blocks_config is not implemented on the st355 path")

    def apply_diffusion_blocks_trainable_filter(self) -> None:
        return None

    def get_pipeline(self, pipeline_type=None, load_base_model: bool = True):
        """common.py:4448.  Validation pipelines are diffusers objects (out of scope); the family-agnostic denoising loop over this plugin's own forward
        is simpletuner_amd.sampling.flow_match_euler_sample"""
        raise NotImplementedError("diffusers pipelines are not built on the st355 path; use simpletuner_amd.sampling.flow_match_euler_sample")

    def configure_gradient_checkpointing(self):
        """trainer.py:3635-3646 + common.py:3604-3636: `gradient_checkpointing` turns recomputation on for the trained component,
        `gradient_checkpointing_interval` (> 1) / `gradient_checkpointing_segment_stride` select the segmented modes, the backend must be the plain
        recompute one.  A family whose engine has no recompute path refuses instead of silently keeping every activation."""
        if not getattr(self.config, "gradient_checkpointing", False):
            return
        comp = self.get_trained_component()
        if comp is None or not hasattr(comp, "enable_gradient_checkpointing") or not hasattr(comp, "_checkpoint_segments"):
            raise NotImplementedError(f"gradient_checkpointing: {self.NAME} has no recompute path on the st355 path (built: Flux, SD3, PixArt-Sigma + its "
                                      f"ControlNet branch, the SDXL / SD1.x UNet); with 288 GB of HBM the step keeps its activations — drop the flag")
        comp.enable_gradient_checkpointing()
        interval = getattr(self.config, "gradient_checkpointing_interval", None)
        if interval is not None and int(interval) > 1:
            comp.set_gradient_checkpointing_interval(int(interval))
        stride = getattr(self.config, "gradient_checkpointing_segment_stride", None)
        if stride is not None:
            comp.set_gradient_checkpointing_segment_stride(int(stride))
        comp.set_gradient_checkpointing_backend(str(getattr(self.config, "gradient_checkpointing_backend", "torch") or "torch"))

    # ---- API parity with the reference plugin surface (common.py:3691-3781) ----
    def fuse_qkv_projections(self):
        """no-op: projections that share an input are ALWAYS stored and executed as one matrix on this path; the per-projection diffusers
        keys are views into it (PackedJointAttnProcessor2_0's fused to_qkv, packed_attention_processors.py:126-187)"""
        return None

    def unfuse_qkv_projections(self):
        return None

    # ---- LoRA checkpoints: the reference's layout (save_hooks.py:850-895 -> pipeline.save_lora_weights) ----
    LORA_WEIGHT_NAME = "pytorch_lora_weights.safetensors"

    def lora_state_dict(self, component=None):
        """peft.get_peft_model_state_dict layout: `<module>.lora_A.weight` / `<module>.lora_B.weight` (the adapter name is dropped)"""
        comp = component if component is not None else self.get_trained_component()
        src = comp.lora_state_dict().items() if hasattr(comp, "lora_state_dict") else ((n, p.detach()) for n, p in comp.named_parameters() if ".lora_" in n)
        # (a component whose working layout pads the adapter factors — PixArt's 72 -> 96 head lanes — hands out the true peft shapes itself)
        return {n.replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B."): p for n, p in src}

    COMFYUI_LORA_PRESERVE_COMPONENT_PREFIXES = None     # common.py:524; Flux / SD3 / PixArt keep their `transformer.` prefix in ComfyUI files
    AUTO_LORA_FORMAT_DETECTION = False                  # flux/model.py:56: a diffusers-configured run still recognises a ComfyUI file on load

    def _lora_adapter_metadata(self) -> dict:
        """what peft records for the adapter (LoraConfig r / lora_alpha): alpha defaults to the rank (common.py:1049-1128)"""
        r = int(getattr(self.config, "lora_rank", 0) or 0)
        alpha = getattr(self.config, "lora_alpha", None)
        return {"r": r, "lora_alpha": float(alpha if alpha is not None else r)}

    def _convert_lora_state_dict_to_comfyui(self, weights: dict, *, adapter_metadata=None, component_adapter_metadata=None) -> dict:
        from .training.lora_keys import convert_diffusers_to_comfyui
        return convert_diffusers_to_comfyui(weights, adapter_metadata=adapter_metadata,
                                            preserve_component_prefixes=self.COMFYUI_LORA_PRESERVE_COMPONENT_PREFIXES)

    def _convert_lora_state_dict_from_comfyui(self, weights: dict, *, target_prefix: str):
        from .training.lora_keys import convert_comfyui_to_diffusers
        return convert_comfyui_to_diffusers(weights, target_prefix=target_prefix)

    def save_lora_weights(self, output_dir, **layers):
        """writes `pytorch_lora_weights.safetensors` with diffusers' component prefix (`transformer.` / `unet.`), as the diffusers pipelines'
        save_lora_weights the reference calls (common.py:2072-2120); `layers` = {"<subfolder>_lora_layers": state} or nothing (= the trained
        component).  `config.lora_format == "comfyui"` converts the keys on the way out (ComfyUI / kohya names + one `.alpha` tensor per module)."""
        import os

        from safetensors.torch import save_file

        from .training.lora_keys import PEFTLoRAFormat, normalize_lora_format
        if not any(k.endswith("_lora_layers") for k in layers):
            layers = dict(layers, **{f"{self.MODEL_SUBFOLDER}_lora_layers": self.lora_state_dict()})
        flat, meta, comp_meta = {}, {}, {}
        for key, state in layers.items():
            if key.endswith("lora_adapter_metadata") and isinstance(state, dict):
                comp_meta[key[:-len("_lora_adapter_metadata")]] = state
                meta.update(state)
                continue
            if state is None or not key.endswith("_lora_layers"):
                continue
            prefix = key[:-len("_lora_layers")]
            for k, v in state.items():
                flat[f"{prefix}.{k}"] = v.detach().to("cpu").contiguous()
        fmt = normalize_lora_format(getattr(self.config, "lora_format", None))
        if fmt == PEFTLoRAFormat.COMFYUI and not getattr(self, "NATIVE_COMFYUI_LORA_SUPPORT", False):
            flat = self._convert_lora_state_dict_to_comfyui(flat, adapter_metadata=meta or self._lora_adapter_metadata(), component_adapter_metadata=comp_meta)
            flat = {k: v.contiguous() for k, v in flat.items()}
        os.makedirs(output_dir, exist_ok=True)
        path = os.path.join(output_dir, self.LORA_WEIGHT_NAME)
        save_file(flat, path, metadata={"format": "pt"})
        return path

    def _kohya_name_map(self) -> dict:
        """kohya module name -> this component's module path, built forwards from the adapters that exist (the underscore-joined kohya spelling
        cannot be split back into a dotted path on its own)"""
        out = {}
        for n, _p in self.get_trained_component().named_parameters():
            if ".lora_A." in n:
                module = n[:n.index(".lora_A.")]
                out["lora_unet_" + module.replace(".processor.", ".").replace(".", "_")] = module
        return out

    def load_lora_weights(self, models=None, input_dir=None):
        """common.py:1875-2047: read the file back into the adapters of the trained component (strict on the adapter keys).  ComfyUI-dialect files
        (configured, or detected when AUTO_LORA_FORMAT_DETECTION) are converted first; a per-module alpha that disagrees with the configured
        adapter scale is an error here — the adapters' alpha/r scale is fixed at construction, loading must not silently change the model."""
        import os

        from safetensors.torch import load_file

        from .training.lora_keys import PEFTLoRAFormat, detect_state_dict_format, normalize_lora_format, to_peft_keys
        comp = self.get_trained_component()
        flat = load_file(os.path.join(input_dir, self.LORA_WEIGHT_NAME))
        prefix = f"{self.MODEL_SUBFOLDER}."
        fmt = normalize_lora_format(getattr(self.config, "lora_format", None))
        if fmt == PEFTLoRAFormat.DIFFUSERS and self.AUTO_LORA_FORMAT_DETECTION and detect_state_dict_format(flat) == PEFTLoRAFormat.COMFYUI:
            fmt = PEFTLoRAFormat.COMFYUI
        if fmt == PEFTLoRAFormat.COMFYUI:
            alphas = {}
            if any(k.startswith("lora_unet_") for k in flat):                 # kohya names (SD / SDXL export)
                back, conv = self._kohya_name_map(), {}
                for k, v in flat.items():
                    name, _, tail = k.partition(".")
                    if name not in back:
                        continue
                    if tail == "alpha":
                        alphas[f"{prefix}{back[name]}.alpha"] = float(v)
                    else:
                        conv[f"{prefix}{back[name]}." + {"lora_down.weight": "lora_A.weight", "lora_up.weight": "lora_B.weight"}[tail]] = v
                flat = conv
            else:
                flat, alphas = self._convert_lora_state_dict_from_comfyui(flat, target_prefix=self.MODEL_SUBFOLDER)
                flat = to_peft_keys(flat)
            want = self._lora_adapter_metadata()["lora_alpha"]
            bad = {k: a for k, a in alphas.items() if want and abs(a - want) > 1e-6}
            if bad:
                k0 = next(iter(bad))
                raise ValueError(f"LoRA file alpha {bad[k0]} for {k0} differs from the configured lora_alpha {want}; set lora_alpha to match the file")
        own = {n.replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B."): p for n, p in comp.named_parameters() if ".lora_" in n}
        missing = [k for k in own if prefix + k not in flat]
        if missing:
            raise KeyError(f"LoRA checkpoint is missing {len(missing)} adapter tensors, e.g. {missing[:3]}")
        with torch.no_grad():
            if hasattr(comp, "load_lora_state_dict"):              # working layout differs from the peft shapes (PixArt's head padding)
                comp.load_lora_state_dict({k: flat[prefix + k] for k in own})
            else:
                for k, p in own.items():
                    p.copy_(flat[prefix + k].to(device=p.device, dtype=p.dtype))
        return comp

    def uses_noise_schedule(self) -> bool:
        return self.PREDICTION_TYPE is not PredictionTypes.FLOW_MATCHING

    def setup_training_noise_schedule(self):
        """common.py:4518-4556: the TRAINING schedule — DDPM coefficients for epsilon / v families; for flow matching the Euler scheduler with the
        static shift, its bounds reset to the unshifted ones (fix_flow_match_euler_schedule_bounds)"""
        if self.PREDICTION_TYPE is not PredictionTypes.FLOW_MATCHING:
            self.noise_schedule = DDPMSchedule(prediction_type=self.PREDICTION_TYPE.value, device=self.accelerator.device,
                                               rescale_betas_zero_snr=bool(getattr(self.config, "rescale_betas_zero_snr", False)))
        else:
            from .sampling import FlowMatchEulerDiscreteScheduler, fix_flow_match_euler_schedule_bounds
            self.noise_schedule = fix_flow_match_euler_schedule_bounds(
                FlowMatchEulerDiscreteScheduler(shift=float(getattr(self.config, "flow_schedule_shift", None) or 1.0)))
        return self.noise_schedule

    def flow_matching_target_direction(self) -> float:
        return 1.0

    # ---- sigma / timestep sampling (common.py:4994-5090) ----
    def sample_flow_sigmas(self, batch: dict, state: dict):
        """common.py:4994-5090: mixflow | custom timestep list (fixed-list / round-robin) | sigmoid-normal (default) | uniform | Beta | the
        "fast" discrete schedule | cubic-spline density (flow_cubic_schedule_weights, training/sigma_density.py)"""
        cfg = self.config
        bsz = batch["latents"].shape[0]
        dev = self.accelerator.device
        shape_ref = batch.get("noise_shape_ref", batch.get("noise", batch["latents"]))
        if getattr(cfg, "mixflow_enabled", False) is True:
            # sigma increases toward noise: the paper's t ~ Beta(2,1) becomes sigma = 1 - sqrt(U) ~ Beta(1,2)   (common.py:5001-5007)
            sigmas = 1.0 - torch.sqrt(torch.rand((bsz,), device=dev))
            sigmas = apply_flow_schedule_shift(cfg, self.noise_schedule, sigmas, shape_ref)
            return sigmas, sigmas * 1000.0
        custom = self._normalize_flow_custom_timesteps(getattr(cfg, "flow_custom_timesteps", None))
        if custom is not None:
            mode = str(getattr(cfg, "flow_timesteps_mode", "fixed-list") or "fixed-list").replace("_", "-")
            if mode not in {"fixed-list", "round-robin"}:
                raise ValueError("flow_timesteps_mode must be either 'fixed-list' or 'round-robin'.")
            if torch.max(custom) <= 1.0:                                   # values <= 1 are sigmas, otherwise timesteps in [0, 1000]
                base_s = custom.clamp(0.0, 1.0); base_t = base_s * 1000.0
            else:
                base_t = custom.clamp(0.0, 1000.0); base_s = (base_t / 1000.0).clamp(0.0, 1.0)
            if base_t.numel() == 1:
                return base_s.expand(bsz), base_t.expand(bsz)
            if mode == "round-robin":
                world = int(getattr(self.accelerator, "num_processes", 1) or 1)
                rank = int(getattr(self.accelerator, "process_index", 0) or 0)
                gather = getattr(self.accelerator, "gather", None)
                if world > 1 and callable(gather):                          # ranks may hold different batch sizes (distributed batch layout)
                    sizes = [int(v) for v in gather(torch.tensor([bsz], device=dev)).reshape(-1).tolist()]
                    global_bsz, offset = sum(sizes), sum(sizes[:rank])
                else:
                    global_bsz, offset = bsz * world, bsz * rank
                if not hasattr(self, "_flow_custom_timestep_cursor"):
                    resume = getattr(self, "_flow_custom_timestep_resume_step", None)
                    done = int(resume if resume is not None else state.get("global_step", 0) or 0)
                    self._flow_custom_timestep_cursor = (done * global_bsz) % base_t.numel()
                    if resume is not None:
                        delattr(self, "_flow_custom_timestep_resume_step")
                cursor = int(self._flow_custom_timestep_cursor)
                idx = (torch.arange(bsz, device=dev) + cursor + offset) % base_t.numel()
                self._flow_custom_timestep_cursor = (cursor + global_bsz) % base_t.numel()
            else:
                idx = torch.randint(0, base_t.numel(), (bsz,), device=dev)
            return base_s[idx], base_t[idx]
        if self._uses_flow_cubic_schedule():                                 # common.py:5058-5061: ahead of every other distribution switch
            sigmas = self._sample_flow_cubic_values(bsz, dev)
            sigmas = apply_flow_schedule_shift(cfg, self.noise_schedule, sigmas, shape_ref)
            return sigmas, sigmas * 1000.0
        if getattr(cfg, "flux_fast_schedule", False) and not (getattr(cfg, "flow_use_beta_schedule", False) or getattr(cfg, "flow_use_uniform_schedule", False)):
            import random
            sigmas = torch.tensor(random.choices([1.0] * 7 + [0.75, 0.5, 0.25], k=bsz), device=dev)      # no schedule shift on this branch
            return sigmas, sigmas * 1000.0
        if getattr(cfg, "flow_use_uniform_schedule", False):
            sigmas = torch.rand((bsz,), device=dev)
        elif getattr(cfg, "flow_use_beta_schedule", False):
            dist = torch.distributions.Beta(cfg.flow_beta_schedule_alpha, cfg.flow_beta_schedule_beta)
            sigmas = dist.sample((bsz,)).to(device=dev)
        else:
            normal = torch.randn((bsz,), device=dev)
            offset = self._get_dataset_timestep_sampling_offset(batch)      # per-dataset bias of the logit-normal mean (common.py:5069-5071)
            if offset:
                normal = normal + offset
            sigmas = torch.sigmoid(getattr(cfg, "flow_sigmoid_scale", 1.0) * normal)
        sigmas = apply_flow_schedule_shift(cfg, self.noise_schedule, sigmas, shape_ref)
        return sigmas, sigmas * 1000.0

    # resolution-dependent shift for schedulers with dynamic shifting (common.py:4730-4797)
    def _get_patch_size_for_dynamic_shift(self, noise_scheduler):
        try:
            comp = self.get_trained_component()
        except Exception:
            comp = None
        for holder in (getattr(comp, "config", None), getattr(noise_scheduler, "config", None), self.config):
            ps = getattr(holder, "patch_size", None) if holder is not None else None
            if ps is not None:
                return ps
        return None

    def calculate_dynamic_shift_mu(self, noise_scheduler, latents):
        """mu(seq_len): the line through (base_image_seq_len, base_shift) and (max_image_seq_len, max_shift); seq_len counts patches (x frames
        for [B,C,F,H,W] latents)"""
        sc = getattr(noise_scheduler, "config", None)
        if latents is None or sc is None:
            return None
        need = ("base_image_seq_len", "max_image_seq_len", "base_shift", "max_shift")
        absent = [f for f in need if getattr(sc, f, None) is None]
        if absent:
            raise ValueError(f"Cannot compute dynamic timestep shift; scheduler is missing config values: {', '.join(absent)}")
        ps = self._get_patch_size_for_dynamic_shift(noise_scheduler)
        if ps is None or ps <= 0:
            raise ValueError("Cannot compute dynamic timestep shift because no valid `patch_size` was found.")
        frames = latents.shape[-3] if latents.ndim == 5 else 1
        seq_len = frames * (int(latents.shape[-2]) // int(ps)) * (int(latents.shape[-1]) // int(ps))
        return calculate_shift(seq_len, sc.base_image_seq_len, sc.max_image_seq_len, sc.base_shift, sc.max_shift)

    # cubic-spline sigma density (common.py:4840-4854)
    def _flow_cubic_schedule_weights(self):
        from .training.sigma_density import parse_cubic_spline_weights
        return parse_cubic_spline_weights(getattr(self.config, "flow_cubic_schedule_weights", None))

    def _uses_flow_cubic_schedule(self) -> bool:
        return self._flow_cubic_schedule_weights() is not None

    def _sample_flow_cubic_values(self, batch_size: int, device) -> torch.Tensor:
        from .training.sigma_density import CubicSplineDistribution
        knots = self._flow_cubic_schedule_weights()
        if knots is None:
            raise ValueError("flow_cubic_schedule_weights must be configured before sampling its distribution.")
        key = (knots, str(device))
        if getattr(self, "_flow_cubic_distribution_cache_key", None) != key:       # the table is built once per (knots, device)
            self._flow_cubic_distribution = CubicSplineDistribution(knots, device=device)
            self._flow_cubic_distribution_cache_key = key
        return self._flow_cubic_distribution.sample((batch_size,))

    def _normalize_flow_custom_timesteps(self, raw):
        """common.py:4799-4838: comma / semicolon separated string, JSON list, or array -> finite 1-D fp32 tensor (None when empty)"""
        import json
        if raw in (None, "", "None"):
            return None
        cand = raw
        if isinstance(cand, str):
            st_ = cand.strip()
            if st_ == "":
                return None
            try:
                cand = json.loads(st_)
            except Exception:
                try:
                    cand = [float(seg.strip()) for seg in st_.replace(";", ",").split(",") if seg.strip()]
                except Exception:
                    return None
        try:
            t = torch.as_tensor(cand, device=self.accelerator.device, dtype=torch.float32).flatten()
        except Exception:
            return None
        t = t[torch.isfinite(t)]
        return t if t.numel() else None

    def reset_flow_custom_timestep_cursor(self, global_step: int = 0) -> None:
        if hasattr(self, "_flow_custom_timestep_cursor"):
            delattr(self, "_flow_custom_timestep_cursor")
        self._flow_custom_timestep_resume_step = int(global_step or 0)

    # round-robin cursor in checkpoints (common.py:4861-4912): one small JSON per rank next to the optimizer state
    def _flow_custom_timestep_state_path(self, ckpt_dir: str) -> str:
        import os
        rank = _process_rank(self.accelerator)
        return os.path.join(ckpt_dir, "flow_custom_timestep_state.json" if rank == 0 else f"flow_custom_timestep_state-{rank}.json")

    def save_flow_custom_timestep_state(self, ckpt_dir: str) -> None:
        import json
        import os
        mode = str(getattr(self.config, "flow_timesteps_mode", "fixed-list") or "fixed-list").replace("_", "-")
        if mode != "round-robin" or self._normalize_flow_custom_timesteps(getattr(self.config, "flow_custom_timesteps", None)) is None:
            return
        state = {"rank": _process_rank(self.accelerator)}
        if hasattr(self, "_flow_custom_timestep_cursor"):
            state["cursor"] = int(self._flow_custom_timestep_cursor)
        elif hasattr(self, "_flow_custom_timestep_resume_step"):
            state["resume_step"] = int(self._flow_custom_timestep_resume_step)
        else:
            return
        os.makedirs(ckpt_dir, exist_ok=True)
        path = self._flow_custom_timestep_state_path(ckpt_dir)
        with open(path + ".tmp", "w", encoding="utf-8") as fh:
            json.dump(state, fh)
        os.replace(path + ".tmp", path)                                     # atomic: a crash never leaves a half-written cursor

    def load_flow_custom_timestep_state(self, ckpt_dir: str, fallback_global_step: int = 0) -> bool:
        import json
        import os
        own = self._flow_custom_timestep_state_path(ckpt_dir)
        shared = os.path.join(ckpt_dir, "flow_custom_timestep_state.json")
        for path in dict.fromkeys((own, shared)):
            if not os.path.exists(path):
                continue
            with open(path, "r", encoding="utf-8") as fh:
                state = json.load(fh)
            if state.get("cursor") is not None:
                self._flow_custom_timestep_cursor = int(state["cursor"])
                if hasattr(self, "_flow_custom_timestep_resume_step"):
                    delattr(self, "_flow_custom_timestep_resume_step")
                return True
            if state.get("resume_step") is not None:
                self.reset_flow_custom_timestep_cursor(int(state["resume_step"]))
                return True
        self.reset_flow_custom_timestep_cursor(fallback_global_step)
        return False

    def _get_dataset_timestep_sampling_offset(self, batch: dict) -> float:
        """common.py:4915-4918: `timestep_sampling_offset` of the dataset the batch came from.  The reference asks its StateTracker; the drop-in
        asks `DATA_BACKEND_CONFIGS` (a plain {backend_id: config} map the data side fills, or a callable `id -> config`)."""
        return float(get_data_backend_config(batch.get("data_backend_id")).get("timestep_sampling_offset", 0.0))

    def _mixflow_gamma(self) -> float:
        gamma = float(getattr(self.config, "mixflow_gamma", 0.8))
        if not 0.0 <= gamma <= 1.0:
            raise ValueError("mixflow_gamma must be between 0.0 and 1.0.")
        return gamma

    def _mixflow_interpolation_sigmas(self, sigmas, slowdown_factors=None):
        """common.py:4966-4977: the INTERPOLATION time is slowed toward noise, the model still sees `sigmas`"""
        sv = sigmas.reshape(sigmas.shape[0], -1)[:, 0]
        gamma = self._mixflow_gamma()
        if gamma == 0.0:
            return sv
        if slowdown_factors is None:
            slowdown_factors = torch.rand_like(sv)
        return sv + slowdown_factors * gamma * (1.0 - sv)

    # ---- prepare_batch (common.py:5862-6041) ----
    def prepare_batch(self, batch: dict, state: dict) -> dict:
        if not batch:
            return batch
        dev, wd = self.accelerator.device, self.config.weight_dtype
        if wd != BF16:
            raise NotImplementedError("the st355 path computes in bf16 (mixed_precision=bf16)")
        if batch.get("prompt_embeds") is not None and hasattr(batch["prompt_embeds"], "to"):
            batch["encoder_hidden_states"] = batch["prompt_embeds"].to(device=dev, dtype=wd)
        pooled = batch.get("add_text_embeds")
        time_ids = batch.get("batch_time_ids")
        batch["added_cond_kwargs"] = {}
        if pooled is not None and hasattr(pooled, "to"):
            batch["added_cond_kwargs"]["text_embeds"] = pooled.to(device=dev, dtype=wd)
        if time_ids is not None and hasattr(time_ids, "to"):
            batch["added_cond_kwargs"]["time_ids"] = time_ids.to(device=dev, dtype=wd)
        latents = batch.get("latent_batch")
        if not hasattr(latents, "to"):
            raise ValueError("Received invalid value for latents.")
        batch["latents"] = latents.to(device=dev, dtype=wd).contiguous()
        mask = batch.get("encoder_attention_mask")
        if mask is not None and hasattr(mask, "to"):
            batch["encoder_attention_mask"] = mask.to(device=dev, dtype=wd)
        perturb = getattr(self.config, "input_perturbation", 0) or 0

        lat = batch["latents"]
        batch["noise_shape_ref"] = lat
        if self.PREDICTION_TYPE is PredictionTypes.FLOW_MATCHING:
            batch["sigmas"], batch["timesteps"] = self.sample_flow_sigmas(batch=batch, state=state)
            sig = batch["sigmas"].to(device=dev, dtype=torch.float32).contiguous()
            if getattr(self.config, "mixflow_enabled", False) is True:          # _prepare_flow_noisy_latents (common.py:4979-4992)
                gamma = self._mixflow_gamma()
                slow = torch.rand_like(sig) if gamma > 0.0 else torch.zeros_like(sig)
                sig = self._mixflow_interpolation_sigmas(sig, slow).contiguous()
                batch["mixflow_slowdown_factors"], batch["mixflow_interpolation_sigmas"] = slow, sig
            given = batch.get("noise")          # tests / parity runs inject the reference's noise; else Philox in-kernel
            seed = int(getattr(self.config, "seed", 42) or 0) + 1_000_003 * int(getattr(self.accelerator, "process_index", 0))
            per_call = (lat.numel() + 3) // 4
            noisy, target, noise = ops.flow_noise_mix(lat, sig, noise=given, seed=seed, offset=self._noise_offset)
            self._noise_step += 1
            self._noise_offset += per_call          # mixed aspect buckets / varying batch: ranges of different steps never overlap
            input_noise = noise
            steps_ = getattr(self.config, "input_perturbation_steps", None)
            if perturb != 0 and (not steps_ or state.get("global_step", 0) < steps_):
                # common.py:5957-5968 + _prepare_flow_noisy_latents (:4975-4992): the INPUT noise is perturbed — x_t = (1 - sigma) x + sigma (n + p e) — while the target
                # keeps the un-perturbed n - x (get_prediction_target reads batch["noise"], :4610-4611).  The first pass above produced n and the target; the mix is
                # redone with the perturbed noise (an option off by default: one extra streaming pass when it is on)
                p_ = float(perturb) * ((1.0 - state.get("global_step", 0) / steps_) if steps_ else 1.0)
                input_noise = (noise + p_ * torch.randn_like(lat)).to(noise.dtype)
                noisy, _, _ = ops.flow_noise_mix(lat, sig, noise=input_noise)
            batch["noise"] = noise
            batch["input_noise"] = input_noise
            batch["noisy_latents"] = noisy
            batch["flow_target"] = target      # n - x (common.py:4610-4611), consumed by get_prediction_target
            self.expand_sigmas(batch)
        else:
            # DDPM families (common.py:5983-6002): discrete timesteps (segmented selection when bsz > 1, custom_schedule.py:18-58), noise =
            # randn_like(latents) (or injected), x_t = sqrt(acp_t) x + sqrt(1 - acp_t) n in fp32 then cast (DDPMScheduler.add_noise)
            if self.noise_schedule is None:
                self.setup_training_noise_schedule()
            sched = self.noise_schedule
            bsz = lat.shape[0]
            given_t = batch.get("timesteps")
            if given_t is None:
                given_t = sched.sample_timesteps(bsz, segmented=not getattr(self.config, "disable_segmented_timestep_sampling", False),
                                                 weights=DDPMSchedule.timestep_weights(self.config, sched.config.num_train_timesteps),
                                                 refiner_training=bool(getattr(self.config, "refiner_training", False)),
                                                 refiner_invert_schedule=bool(getattr(self.config, "refiner_training_invert_schedule", False)),
                                                 refiner_strength=float(getattr(self.config, "refiner_training_strength", 0.2)))
            batch["timesteps"] = given_t.to(device=dev).long()
            noise = batch.get("noise")
            if noise is None:
                noise = torch.randn_like(lat)
                if getattr(self.config, "offset_noise", False):                       # common.py:5940-5949 ([B,C,1,1] offset, drawn with a probability)
                    import random
                    prob = float(getattr(self.config, "noise_offset_probability", 1.0))
                    if prob == 1.0 or random.random() < prob:
                        noise = noise + float(getattr(self.config, "noise_offset", 0.1)) * torch.randn(lat.shape[0], lat.shape[1], 1, 1, device=dev, dtype=lat.dtype)
            noise = noise.to(device=dev, dtype=wd)
            input_noise = noise
            steps_ = getattr(self.config, "input_perturbation_steps", None)
            if perturb != 0 and (not steps_ or state.get("global_step", 0) < steps_):     # common.py:5957-5968: perturb the INPUT noise only
                p_ = float(perturb) * ((1.0 - state.get("global_step", 0) / steps_) if steps_ else 1.0)
                input_noise = noise + p_ * torch.randn_like(lat)
            a, b = sched.mix_coefficients(batch["timesteps"])
            noisy, _ = ops.ddpm_noise_mix(lat, input_noise, a, b, want_v=False)
            batch["noise"] = noise
            batch["input_noise"] = input_noise
            batch["noisy_latents"] = noisy
            if self.PREDICTION_TYPE is PredictionTypes.V_PREDICTION:                  # get_velocity uses the un-perturbed noise (common.py:4649-4653)
                batch["velocity_target"] = ops.ddpm_noise_mix(lat, noise, a, b, want_v=True)[1]
        batch.pop("noise_shape_ref", None)
        return self.prepare_batch_conditions(batch=batch, state=state)

    def expand_sigmas(self, batch: dict) -> dict:
        s = batch["sigmas"]
        batch["sigmas"] = s.reshape(s.shape[0], *([1] * (batch["latents"].dim() - 1)))
        return batch

    def prepare_batch_conditions(self, batch: dict, state: dict) -> dict:
        return batch

    # ---- targets / loss (common.py:4635-4658, 6217-6430) ----
    def get_prediction_target(self, prepared_batch: dict):
        if prepared_batch.get("target") is not None:
            return prepared_batch["target"]
        if self.PREDICTION_TYPE is PredictionTypes.FLOW_MATCHING:
            t = prepared_batch.get("flow_target")
            if t is None:
                # (noise - latents), one extra streaming pass only when the fused target was not kept
                _, t, _ = ops.flow_noise_mix(prepared_batch["latents"], torch.zeros(prepared_batch["latents"].shape[0],
                                             device=prepared_batch["latents"].device), noise=prepared_batch["noise"])
            return t
        if self.PREDICTION_TYPE is PredictionTypes.EPSILON:
            return prepared_batch["noise"]
        if self.PREDICTION_TYPE is PredictionTypes.V_PREDICTION:
            return prepared_batch["velocity_target"]        # noise_schedule.get_velocity (common.py:4649-4653), fused into the noising pass
        if self.PREDICTION_TYPE is PredictionTypes.SAMPLE:
            return prepared_batch["latents"]
        raise ValueError(f"Unknown prediction type {self.PREDICTION_TYPE}.")

    def loss(self, prepared_batch: dict, model_output, apply_conditioning_mask: bool = True, _per_sample_only: bool = False, _row_weight=None):
        """common.py:6217-6430.  The two private switches serve XM (xm.py): `_per_sample_only` returns (weighted per-row losses [B] fp32, the row
        weights or None) without touching autograd; `_row_weight` replaces the row weights (it already carries the min-SNR factor)."""
        target = self.get_prediction_target(prepared_batch)
        model_pred = model_output["model_prediction"]
        if target is None:
            raise ValueError("Target is None. Cannot compute loss.")
        loss_type = getattr(self.config, "loss_type", "l2")
        if loss_type not in ("l2", "huber", "smooth_l1"):
            raise NotImplementedError(f"Unsupported Loss Type {loss_type}")
        emask = self._conditioning_loss_mask(prepared_batch, model_pred, apply_conditioning_mask)      # common.py:6402-6424
        weight = None
        if self.PREDICTION_TYPE in (PredictionTypes.EPSILON, PredictionTypes.V_PREDICTION):
            gamma = getattr(self.config, "snr_gamma", None)
            if gamma is not None and gamma > 0:                                    # min-SNR weighting (common.py:6363-6397)
                weight = self.noise_schedule.min_snr_weights(prepared_batch["timesteps"], gamma, self.PREDICTION_TYPE is PredictionTypes.V_PREDICTION)
                weight = weight.to(device=model_pred.device, dtype=torch.float32).contiguous()
            elif loss_type == "l2" and float(getattr(self.config, "snr_weight", 1.0)) != 1.0:
                weight = torch.full((model_pred.shape[0],), float(self.config.snr_weight), dtype=torch.float32, device=model_pred.device)
        if _row_weight is not None:
            weight = _row_weight.to(device=model_pred.device, dtype=torch.float32).contiguous()
        huber_c = None
        if loss_type != "l2":
            # huber / smooth_l1 (common.py:6248-6281): one huber_c per sample — constant, or scheduled on the timestep (common.py:6168-6216)
            huber_c = self.compute_scheduled_huber_c(prepared_batch["timesteps"]).to(device=model_pred.device, dtype=torch.float32)
            huber_c = huber_c.reshape(-1).expand(model_pred.shape[0]).contiguous()
        if _per_sample_only:
            with torch.no_grad():
                p_, t_ = model_pred.detach().to(BF16), target.to(BF16)
                per = ops.cond_loss(p_, t_, loss_type, huber_c, weight=weight, want_grad=False, emask=emask)[1]
            return per, weight
        if loss_type == "l2" and emask is None:
            return _MSELossFn.apply(model_pred, target, weight)
        return _CondLossFn.apply(model_pred, target, loss_type, huber_c, weight, emask)

    def _conditioning_loss_mask(self, prepared_batch: dict, model_pred, apply_conditioning_mask: bool):
        """common.py:6402-6424: `loss_mask_type` (legacy: `conditioning_type`) "mask" multiplies the elementwise loss by the first channel of
        `conditioning_pixel_values`, area-resized to the latent grid and mapped from [-1,1] to [0,1]; "segmentation" — with probability
        `masked_loss_probability` — by the binarised channel mean.  Returns the fp32 [B, H*W] element mask the fused loss kernel broadcasts over the
        channels, or None.  (The area resize of a one-channel image is input preparation, done with torch like the reference does.)"""
        kind = prepared_batch.get("loss_mask_type")
        if not kind:
            legacy = prepared_batch.get("conditioning_type")
            kind = legacy if legacy in ("mask", "segmentation") else None
        if kind not in ("mask", "segmentation") or not apply_conditioning_mask:
            return None
        if model_pred.dim() != 4:
            raise NotImplementedError("conditioning-mask losses are built for [B,C,H,W] predictions")
        cpv = prepared_batch["conditioning_pixel_values"].to(device=model_pred.device, dtype=torch.float32)
        if kind == "mask":
            m = cpv[:, 0].unsqueeze(1)
        else:
            import random
            if not random.random() < float(getattr(self.config, "masked_loss_probability", 1.0)):
                return None
            m = torch.sum(cpv, dim=1, keepdim=True) / 3
        m = torch.nn.functional.interpolate(m, size=model_pred.shape[2:], mode="area") / 2 + 0.5
        if kind == "segmentation":
            m = (m > 0).to(torch.float32)
        return m.reshape(m.shape[0], -1).contiguous()

    def compute_scheduled_huber_c(self, timesteps: torch.Tensor) -> torch.Tensor:
        """common.py:6168-6216 (flow-matching branch of the "snr" schedule: sigma = ((1 - t/1000) / (t/1000 + 1e-4))^0.5)"""
        import math
        schedule = getattr(self.config, "huber_schedule", "constant")
        base = float(getattr(self.config, "huber_c", 0.1))
        t = timesteps.to(torch.float32)
        if schedule == "constant":
            return torch.full_like(t, base)
        if schedule == "exponential":
            sched_cfg = getattr(getattr(self, "noise_schedule", None), "config", None)          # common.py:6187: self.noise_schedule.config.num_train_timesteps
            alpha = -math.log(base) / float(getattr(sched_cfg, "num_train_timesteps", getattr(self.config, "num_train_timesteps", 1000)))
            return torch.exp(-alpha * t)
        if schedule == "snr":
            if self.PREDICTION_TYPE != PredictionTypes.FLOW_MATCHING:      # DDPM: sigma = sqrt((1 - acp_t) / acp_t) (common.py:6199-6205)
                a = self.noise_schedule.acp_on(timesteps.device)[timesteps.long()].float()
                s_ = ((1.0 - a) / a) ** 0.5
                return (1 - base) / (1 + s_) ** 2 + base
            s = t / 1000
            s = ((1.0 - s) / (s + 0.0001)) ** 0.5
            return (1 - base) / (1 + s) ** 2 + base
        raise NotImplementedError(f"Unknown Huber loss schedule {schedule}")

    def loss_with_logs(self, prepared_batch: dict, model_output, apply_conditioning_mask: bool = True):
        """xm_mixin.py:476-485"""
        k = model_output.get("xm_candidate_count") if isinstance(model_output, dict) else None
        if k:
            return self._xm_noise_loss_with_logs(prepared_batch, model_output, candidate_count=int(k), apply_conditioning_mask=apply_conditioning_mask)
        return self.loss(prepared_batch, model_output, apply_conditioning_mask), None

    def auxiliary_loss(self, model_output, prepared_batch: dict, loss: torch.Tensor):
        return loss, None


class _CondLossFn(torch.autograd.Function):
    """huber / smooth_l1 (conditional_loss) -> per-sample mean -> batch mean, gradient from the same kernel pass"""

    @staticmethod
    def forward(ctx, pred, target, loss_type, huber_c, weight=None, emask=None):
        loss, _per, dpred = ops.cond_loss(pred.to(BF16), target.to(BF16), loss_type, huber_c, weight=weight, want_grad=True, emask=emask)
        ctx.save_for_backward(dpred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return (dpred.float() * g).to(dpred.dtype) if g.numel() == 1 else dpred, None, None, None, None, None


class _MSELossFn(torch.autograd.Function):
    """mean_b(mean_chw((pred - target)^2)) in fp32 with the gradient produced by the same kernel pass (K13)."""

    @staticmethod
    def forward(ctx, pred, target, weight=None):
        loss, _per, dpred = ops.mse_loss(pred.to(BF16), target.to(BF16), weight=weight, want_grad=True)
        ctx.save_for_backward(dpred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        # g is the upstream scalar (1.0, or the loss scale); fold it without a host sync
        return (dpred.float() * g).to(dpred.dtype) if g.numel() == 1 else dpred, None, None
