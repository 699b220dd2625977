"""CPU restatement of the step's non-network arithmetic.  TEST INFRASTRUCTURE (see oracle/__init__.py).

    flow noising / target           simpletuner/helpers/models/common.py:4975-4992, 4610-4611
    flow schedule shift             simpletuner/helpers/training/custom_schedule.py:443-478
    sigma sampling (sigmoid-normal) simpletuner/helpers/models/common.py:5062-5068
    MSE loss (per-sample mean)      simpletuner/helpers/models/common.py:6286, 6426-6429
    torch.optim.AdamW               entry at simpletuner/helpers/training/optimizer_param.py:87-96 (math is torch's)
    EMA decay + update              simpletuner/helpers/training/ema.py:322-349, 393-433
PINNED: shift / EMA decay / EMA update / noising against tests/golden/reference_vectors.pt (reference code executed here);
AdamW against torch.optim.AdamW itself.
"""
from __future__ import annotations

import torch


def flow_noisy_and_target(latents, noise, sigmas):
    """x_t = (1 - sigma) x + sigma n ; target = n - x  (flow_matching_target_direction = +1)"""
    s = sigmas.reshape(-1, *([1] * (latents.dim() - 1))).to(latents.dtype)
    return (1.0 - s) * latents + s * noise, noise - latents


def apply_flow_schedule_shift(sigmas, shift):
    if shift is not None and shift > 0:
        return (sigmas * shift) / (1 + (shift - 1) * sigmas)
    return sigmas


def sample_sigmas_sigmoid_normal(normal, sigmoid_scale: float = 1.0, shift=None):
    return apply_flow_schedule_shift(torch.sigmoid(sigmoid_scale * normal), shift)


def mse_loss(pred, target, weight=None):
    l = (pred.float() - target.float()) ** 2
    if weight is not None:
        l = l * weight.reshape(-1, *([1] * (l.dim() - 1)))
    per = l.mean(dim=list(range(1, l.dim())))
    return per.mean(), per


def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    """torch.optim.AdamW single-tensor semantics (decoupled decay; bias-corrected)."""
    p = p * (1 - lr * weight_decay)
    m = m + (g - m) * (1 - beta1)
    v = v * beta2 + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def ema_get_decay(optimization_step, decay, min_decay=0.0, update_after_step=0, warmup_steps=0, use_ema_warmup=False, inv_gamma=1.0,
                  power=2 / 3):
    """ema.py:322-349"""
    step = max(0, optimization_step - update_after_step - 1)
    if warmup_steps > 0:
        if optimization_step < warmup_steps:
            return 0.0
        return decay
    if step <= 0:
        return 0.0
    if use_ema_warmup:
        cur = 1 - (1 + step / inv_gamma) ** -power
    else:
        cur = (1 + step) / (10 + step)
    cur = min(cur, decay)
    return max(cur, min_decay)


def ema_update(shadow, param, decay):
    """ema.py:423: s -= (1 - d) (s - p)"""
    return shadow - (1 - decay) * (shadow - param)
