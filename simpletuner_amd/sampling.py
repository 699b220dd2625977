"""Flow-matching Euler schedule + the sampling loop built on it — SURVEY.md §8(c) `tests/test_flow_match_scheduler_bounds.py` and §8(f)4.

`FlowMatchEulerDiscreteScheduler` restates the scheduler the reference loads as its flow-matching *training* schedule (common.py:4528-4536:
`diffusers.FlowMatchEulerDiscreteScheduler(shift=flow_schedule_shift)` followed by `fix_flow_match_euler_schedule_bounds`) and steps its
validation pipelines with.  diffusers is un-vendored; the arithmetic is corroborated in-tree by the vendored copy at
simpletuner/helpers/models/ace_step/schedulers/scheduling_flow_match_euler_discrete.py:71-330, which tools/gen_golden.py executes (with its
three diffusers imports shimmed) to produce tests/golden/flow_match_scheduler_vectors.pt.

The one behavioural subtlety the reference tests pin: with a STATIC shift, upstream computes `sigma_min / sigma_max` from the already-shifted
training sigmas, so `set_timesteps` — which spaces timesteps between those bounds and shifts again — applies the shift twice
(test_flow_match_scheduler_bounds.py:14-22 keeps that regression visible).  `fix_flow_match_euler_schedule_bounds` (training/flow_match.py:8-19)
resets the bounds to the unshifted 1/N and 1.0; the vendored ACE-Step copy takes the bounds before shifting and needs no fix.  Both forms
are here: `bounds="shifted"` (upstream, the default, to be passed through the fix exactly as the reference does) and `bounds="unshifted"`.

Host logic: a few hundred scalars.  The per-step tensor update `x += (sigma_next - sigma) * v` is one fused torch op on whatever device the
latents live on; the model call inside the loop is the same HIP forward the train step uses.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch


def fix_flow_match_euler_schedule_bounds(scheduler):
    """training/flow_match.py:8-19: static-shift schedules get sigma_max = config.sigma_max (1.0) and sigma_min = 1 / num_train_timesteps back"""
    cfg = getattr(scheduler, "config", None)
    if cfg is None or getattr(cfg, "use_dynamic_shifting", False):
        return scheduler
    n = getattr(cfg, "num_train_timesteps", None)
    if n is None:
        return scheduler
    scheduler.sigma_max = float(getattr(cfg, "sigma_max", 1.0) or 1.0)
    scheduler.sigma_min = 1.0 / float(n)
    return scheduler


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, use_dynamic_shifting: bool = False, base_shift: Optional[float] = 0.5,
                 max_shift: Optional[float] = 1.15, base_image_seq_len: Optional[int] = 256, max_image_seq_len: Optional[int] = 4096,
                 sigma_max: Optional[float] = 1.0, bounds: str = "shifted"):
        if bounds not in ("shifted", "unshifted"):
            raise ValueError("bounds must be 'shifted' (upstream diffusers) or 'unshifted' (the vendored ACE-Step form)")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=use_dynamic_shifting, base_shift=base_shift,
                                      max_shift=max_shift, base_image_seq_len=base_image_seq_len, max_image_seq_len=max_image_seq_len, sigma_max=sigma_max)
        t = np.linspace(1.0, (sigma_max or 1.0) * num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(t).to(torch.float32) / num_train_timesteps
        raw_min, raw_max = sig[-1].item(), sig[0].item()
        if not use_dynamic_shifting:                                   # dynamic: shifted on the fly from the image resolution (set_timesteps(mu=))
            sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig.to("cpu")
        if bounds == "unshifted":
            self.sigma_min, self.sigma_max = raw_min, raw_max
        else:
            self.sigma_min, self.sigma_max = self.sigmas[-1].item(), self.sigmas[0].item()
        self.num_inference_steps = None
        self._step_index = None
        self._begin_index = None

    step_index = property(lambda self: self._step_index)
    begin_index = property(lambda self: self._begin_index)

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def _sigma_to_t(self, sigma):
        return sigma * self.config.num_train_timesteps

    @staticmethod
    def time_shift(mu: float, sigma: float, t):
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, sigmas: Optional[Sequence[float]] = None, mu: Optional[float] = None):
        """simpletuner/helpers/models/ace_step/schedulers/scheduling_flow_match_euler_discrete.py:196-240"""
        if self.config.use_dynamic_shifting and mu is None:
            raise ValueError(" you have a pass a value for `mu` when `use_dynamic_shifting` is set to be `True`")
        if sigmas is None:
            self.num_inference_steps = num_inference_steps
            sigmas = np.linspace(self._sigma_to_t(self.sigma_max), self._sigma_to_t(self.sigma_min), num_inference_steps) / self.config.num_train_timesteps
        else:
            sigmas = np.asarray(sigmas, dtype=np.float64)
            self.num_inference_steps = len(sigmas)
        if self.config.use_dynamic_shifting:
            sigmas = self.time_shift(mu, 1.0, sigmas)
        else:
            sigmas = self.config.shift * sigmas / (1 + (self.config.shift - 1) * sigmas)
        sigmas = torch.from_numpy(np.asarray(sigmas)).to(dtype=torch.float32, device=device)
        self.timesteps = (sigmas * self.config.num_train_timesteps).to(device=device)
        self.sigmas = torch.cat([sigmas, torch.zeros(1, device=sigmas.device)])
        self._step_index = None
        self._begin_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        sched = self.timesteps if schedule_timesteps is None else schedule_timesteps
        hits = (sched == timestep).nonzero()
        return hits[1 if len(hits) > 1 else 0].item()                  # a duplicated first timestep resolves to its second entry (img2img starts)

    def _init_step_index(self, timestep):
        if self._begin_index is None:
            if isinstance(timestep, torch.Tensor):
                timestep = timestep.to(self.timesteps.device)
            self._step_index = self.index_for_timestep(timestep)
        else:
            self._step_index = self._begin_index

    def scale_noise(self, sample: torch.Tensor, timestep: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """forward process at scheduler timesteps: sigma * noise + (1 - sigma) * sample (:128-186)"""
        sig = self.sigmas.to(device=sample.device, dtype=sample.dtype)
        sched = self.timesteps.to(sample.device)
        timestep = timestep.to(sample.device)
        if self._begin_index is None:
            idx = [self.index_for_timestep(t, sched) for t in timestep]
        elif self._step_index is not None:
            idx = [self._step_index] * timestep.shape[0]
        else:
            idx = [self._begin_index] * timestep.shape[0]
        s = sig[idx].flatten()
        s = s.reshape(-1, *([1] * (sample.dim() - 1)))
        return s * noise + (1.0 - s) * sample

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True, **_unused):
        """x_{i+1} = x_i + (sigma_{i+1} - sigma_i) * v, accumulated in fp32, returned in the model output's dtype (:258-330 with omega = 0)"""
        if isinstance(timestep, int) or (torch.is_tensor(timestep) and not timestep.is_floating_point()):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to `EulerDiscreteScheduler.step()` is not supported."
                             " Make sure to pass one of the `scheduler.timesteps` as a timestep.")
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma, sigma_next = self.sigmas[self._step_index], self.sigmas[self._step_index + 1]
        prev = (sample.to(torch.float32) + (sigma_next - sigma) * model_output).to(model_output.dtype)
        self._step_index += 1
        return SimpleNamespace(prev_sample=prev) if return_dict else (prev,)

    def __len__(self):
        return self.config.num_train_timesteps


def flow_match_euler_sample(predict: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], latents: torch.Tensor, scheduler: FlowMatchEulerDiscreteScheduler,
                            num_inference_steps: int, mu: Optional[float] = None, sigmas: Optional[Sequence[float]] = None,
                            on_step: Optional[Callable[[int, torch.Tensor], None]] = None) -> torch.Tensor:
    """The denoising loop of the flow-matching pipelines (e.g. flux/pipeline.py `__call__`): start from noise, call `predict(x, t[B])` — the
    velocity prediction of the trained component, timesteps in scheduler units (0-1000) — and take one Euler step per schedule entry.
    Conditioning, guidance and packing belong to `predict` (the plugin's own forward), so the loop is family-agnostic."""
    scheduler.set_timesteps(num_inference_steps, device=latents.device, mu=mu, sigmas=sigmas)
    x = latents
    with torch.no_grad():
        for i, t in enumerate(scheduler.timesteps):
            v = predict(x, t.expand(x.shape[0]))
            x = scheduler.step(v, t, x, return_dict=False)[0]
            if on_step is not None:
                on_step(i, x)
    return x


class DDIMScheduler:
    """The reference's DEFAULT validation scheduler of its epsilon / v-prediction families (`DEFAULT_NOISE_SCHEDULER = "ddim"`: sdxl/model.py:55,
    sd1x/model.py:45, pixart/model.py:57; `SCHEDULER_NAME_MAP`, training/validation.py:88-102) — diffusers' DDIMScheduler, deterministic form (eta = 0),
    with the Stable Diffusion scheduler_config defaults (scaled_linear betas 0.00085-0.012, 1000 steps, `timestep_spacing="leading"`, `steps_offset=1`,
    `clip_sample=False`, `set_alpha_to_one=False`).  diffusers is un-vendored and no copy of this class exists in the reference tree: the update is the
    published DDIM rule (Song et al., eq. 12 with sigma = 0),
        x0 = (x_t - sqrt(1 - a_t) eps) / sqrt(a_t),      x_{t'} = sqrt(a_{t'}) x0 + sqrt(1 - a_{t'}) eps,
    pinned here by its exact-recovery property (with the true eps of x_t = sqrt(a_t) x0 + sqrt(1 - a_t) eps every step lands on the same x0 / eps pair:
    tests/test_sampling_cpu.py) and sharing its alpha-bar table with the training-side DDPMSchedule (foundation.py), which IS pinned to reference code."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012, beta_schedule: str = "scaled_linear",
                 clip_sample: bool = False, set_alpha_to_one: bool = False, steps_offset: int = 1, prediction_type: str = "epsilon",
                 timestep_spacing: str = "leading", rescale_betas_zero_snr: bool = False):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for DDIMScheduler(st355)")
        if rescale_betas_zero_snr:
            from .foundation import enforce_zero_terminal_snr
            betas = enforce_zero_terminal_snr(betas)
        if prediction_type not in ("epsilon", "v_prediction", "sample"):
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
        if clip_sample:
            raise NotImplementedError("clip_sample is a pixel-space option; latent models run with clip_sample=False")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
                                      clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, prediction_type=prediction_type,
                                      timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.config.train_timesteps`: {T}")
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "leading":
            ts = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        elif sp == "trailing":
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        else:
            raise ValueError(f"{sp} is not supported. Please make sure to choose one of 'leading' or 'trailing'.")
        self.timesteps = torch.from_numpy(ts).to(device)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0, return_dict: bool = True, **_unused):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if eta != 0.0:
            raise NotImplementedError("DDIMScheduler(st355): the deterministic form (eta = 0) is built")
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t].to(torch.float32).item()
        a_prev = (self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod).to(torch.float32).item()
        x, out = sample.to(torch.float32), model_output.to(torch.float32)
        pt = self.config.prediction_type
        if pt == "epsilon":
            x0 = (x - (1 - a_t) ** 0.5 * out) / a_t ** 0.5
            eps = out
        elif pt == "sample":
            x0 = out
            eps = (x - a_t ** 0.5 * x0) / (1 - a_t) ** 0.5
        else:                                                           # v_prediction
            x0 = a_t ** 0.5 * x - (1 - a_t) ** 0.5 * out
            eps = a_t ** 0.5 * out + (1 - a_t) ** 0.5 * x
        prev = (a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps).to(model_output.dtype)
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0.to(model_output.dtype)) if return_dict else (prev,)

    def __len__(self):
        return self.config.num_train_timesteps


def ddim_sample(predict: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], latents: torch.Tensor, scheduler: DDIMScheduler, num_inference_steps: int,
                on_step: Optional[Callable[[int, torch.Tensor], None]] = None) -> torch.Tensor:
    """the denoising loop of the epsilon / v pipelines (sdxl/pipeline.py `__call__` :592+, sd1x/pipeline.py): x_T ~ N(0, I) * init_noise_sigma, one
    scheduler step per entry of `scheduler.timesteps`, `predict(x, t[B])` with integer training timesteps"""
    scheduler.set_timesteps(num_inference_steps, device=latents.device)
    x = latents * scheduler.init_noise_sigma
    with torch.no_grad():
        for i, t in enumerate(scheduler.timesteps):
            out = predict(scheduler.scale_model_input(x, t), t.expand(x.shape[0]))
            x = scheduler.step(out, t, x, return_dict=False)[0]
            if on_step is not None:
                on_step(i, x)
    return x


def cfg_combine(pred_pair: torch.Tensor, guidance_scale: float) -> torch.Tensor:
    """classifier-free guidance on a [2B, ...] prediction of the batch [negative ; positive] (sd3/pipeline.py:1769-1785, sdxl/pipeline.py):
    uncond + g * (text - uncond), combined in fp32"""
    u, c = pred_pair.float().chunk(2)
    return (u + guidance_scale * (c - u)).to(pred_pair.dtype)


def sample_images(plugin, prompt_embeds: torch.Tensor, pooled: Optional[torch.Tensor], latent_height: int, latent_width: int, num_inference_steps: int = 20,
                  generator: Optional[torch.Generator] = None, scheduler=None, mu: Optional[float] = None,
                  decode: bool = True, extra_batch: Optional[dict] = None, guidance_scale: float = 1.0, negative_prompt_embeds: Optional[torch.Tensor] = None,
                  negative_pooled: Optional[torch.Tensor] = None, negative_extra_batch: Optional[dict] = None, latents: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The validation sampling loop end to end on the st355 kernels (training/validation.py -> <family>/pipeline.py `__call__`):
    Gaussian latents -> `num_inference_steps` steps over the plugin's OWN forward (`model_predict`, so packing / ids / guidance / timestep conventions are the
    family's) -> `vae.decode(z / scaling_factor + shift_factor)`.  The scheduler follows the family's prediction type: flow matching -> Euler
    (FlowMatchEulerDiscreteScheduler), epsilon / v-prediction -> DDIM (the reference's default, DEFAULT_NOISE_SCHEDULER).  Classifier-free guidance
    (`guidance_scale` > 1 with negative embeddings): every step runs the model ONCE on the batch [negative ; positive] and combines the halves
    (sd3/pipeline.py:1769-1785).  Returns pixels [B, 3, 8h, 8w] in [-1, 1] (bf16), or the final latents with decode=False.  IP adapters, skip-layer guidance,
    CFG-zero* and the image post-processing of the diffusers pipelines stay outside this tier."""
    dev = plugin.accelerator.device
    B = prompt_embeds.shape[0]
    C = int(plugin.LATENT_CHANNEL_COUNT)
    x = latents if latents is not None else torch.randn(B, C, latent_height, latent_width, device=dev, dtype=torch.float32, generator=generator)
    x = x.to(device=dev, dtype=torch.bfloat16)
    from .foundation import PredictionTypes
    flow = plugin.PREDICTION_TYPE is PredictionTypes.FLOW_MATCHING
    if scheduler is None:
        if flow:
            scheduler = FlowMatchEulerDiscreteScheduler(shift=float(getattr(plugin.config, "flow_schedule_shift", 3.0) or 1.0))
        else:
            # validation.py:2885-2897, 3005-3011: the validation scheduler is rebuilt from the model's scheduler config with the trainer's
            # `inference_scheduler_timestep_spacing` (default "trailing": the grid that visits t = 999, which zero-terminal-SNR models need), `prediction_type`
            # and `rescale_betas_zero_snr` overriding it
            cfg_pt = getattr(plugin.config, "prediction_type", None)
            pt = cfg_pt if cfg_pt in ("epsilon", "v_prediction", "sample") else ("v_prediction" if plugin.PREDICTION_TYPE is PredictionTypes.V_PREDICTION else "epsilon")
            scheduler = DDIMScheduler(prediction_type=pt, timestep_spacing=getattr(plugin.config, "inference_scheduler_timestep_spacing", None) or "trailing",
                                      rescale_betas_zero_snr=bool(getattr(plugin.config, "rescale_betas_zero_snr", False)))
    do_cfg = guidance_scale is not None and guidance_scale > 1.0 and negative_prompt_embeds is not None
    bf = lambda t: None if t is None else t.to(device=dev, dtype=torch.bfloat16)
    pe, pp = bf(prompt_embeds), bf(pooled)
    if do_cfg:                                                        # [negative ; positive], as the pipelines concatenate them
        pe = torch.cat([bf(negative_prompt_embeds), pe], dim=0)
        pp = None if pp is None else torch.cat([bf(negative_pooled if negative_pooled is not None else torch.zeros_like(pooled)), pp], dim=0)

    def pair(v_pos, v_neg):
        if not torch.is_tensor(v_pos):
            return v_pos
        v_neg = v_pos if v_neg is None else v_neg
        return torch.cat([v_neg.to(v_pos.device), v_pos], dim=0)

    def predict(xt, t):
        xin, tin = (torch.cat([xt, xt], dim=0), torch.cat([t, t], dim=0)) if do_cfg else (xt, t)
        batch = {"latents": xin, "noisy_latents": xin, "timesteps": tin.to(device=dev, dtype=torch.float32), "prompt_embeds": pe, "encoder_hidden_states": pe,
                 "add_text_embeds": pp, "added_cond_kwargs": {"text_embeds": pp}}
        if extra_batch:
            neg = negative_extra_batch or {}
            for k, v in extra_batch.items():
                if isinstance(v, dict):                               # e.g. added_cond_kwargs: merged into the entry built above (keeps text_embeds)
                    merged = dict(batch.get(k) or {})
                    merged.update({kk: (pair(vv, (neg.get(k) or {}).get(kk)) if do_cfg else vv) for kk, vv in v.items()})
                    batch[k] = merged
                else:
                    batch[k] = pair(v, neg.get(k)) if do_cfg else v
        out = plugin.model_predict(batch)["model_prediction"].to(xt.dtype)
        return cfg_combine(out, float(guidance_scale)) if do_cfg else out

    if flow:
        x = flow_match_euler_sample(predict, x, scheduler, num_inference_steps, mu=mu)
    else:
        x = ddim_sample(predict, x, scheduler, num_inference_steps)
    if not decode:
        return x
    return plugin.get_vae().decode_scaled(x)
