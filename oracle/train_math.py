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

import math

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


# ------------------------------------------------------------------------------------------------
# AdamWBF16 — the examples' default optimizer (all-bf16 state, stochastic rounding, compensated summation)
# ------------------------------------------------------------------------------------------------
def stochastic_round_bf16(x_f32: torch.Tensor, rand16: torch.Tensor) -> torch.Tensor:
    """copy_stochastic_ (optimizers/adamw_bfloat16/stochastic/__init__.py:47-72): add a random 16-bit integer to the fp32 bit pattern,
    clear the low 16 bits, keep the high half as bf16.  `rand16` int32 in [0, 65536)."""
    bits = (x_f32.contiguous().view(torch.int32) + rand16.to(torch.int32)) & -65536
    return bits.view(torch.float32).to(torch.bfloat16)


def adamw_bf16_decay_schedule(accumulated_decay: float, weight_decay: float, lr: float, threshold: float = 5e-3):
    """optimizers/adamw_bfloat16/__init__.py:92-95: the decay is only APPLIED once the owed amount exceeds the threshold.
    Returns (decay_this_iteration, new_accumulated_decay)."""
    acc = accumulated_decay + weight_decay * lr
    dec = acc if acc > threshold else 0.0
    return dec, acc - dec


def adamw_bf16_step(p, g, m, v, shift, step: int, lr: float, beta1: float, beta2: float, eps: float, decay_this_iteration: float, draws,
                    decay_alpha: str = "fp32"):
    """_make_step (optimizers/adamw_bfloat16/__init__.py:113-180) with the four stochastic-rounding draws injected.  All state bf16.
    Note the reference's add_stochastic_(input, other, alpha) computes  other + alpha * input  (stochastic/__init__.py:90-101), so the
    first moment is  SR(g + (1 - beta1) * (beta1 * m))  — restated as is.  Every intermediate that the reference materialises in bf16 is
    rounded to bf16 here (RNE), every fp32 temporary stays fp32.
    decay_alpha: how the scalar of `shift.add_(p, alpha=-decay)` (a plain bf16 add) is held.  "fp32" = opmath scalar, what ATen's GPU
    kernels do (and what libst355 does); "aten_cpu" = what ATen's CPU kernel did when the golden fixture was generated: the vectorised
    body (32-element chunks) holds alpha as bf16 and computes in fp32, the scalar tail (n % 32 elements) also rounds the product
    to bf16 — quirks of the host library, reproduced only so that the fixture pins this restatement bit for bit."""
    bf, f = torch.bfloat16, torch.float32
    m1 = (m.to(f) * beta1).to(bf)                                        # exp_avg.mul_(beta1)
    m_new = stochastic_round_bf16(g.to(f) + (1 - beta1) * m1.to(f), draws[0])   # add_stochastic_(exp_avg, grad, alpha=1-beta1)
    v1 = (v.to(f) * beta2).to(bf)                                        # exp_avg_sq.mul_(beta2)
    v_new = torch.addcmul(v1.to(f), g.to(f), g.to(f), value=1 - beta2).to(bf)   # .addcmul_(grad, grad, value=1-beta2)
    denom = (v_new.to(f).sqrt().to(bf).to(f) + eps).to(bf)               # exp_avg_sq.sqrt().add_(eps)
    dc = (1 - beta2 ** step) ** 0.5
    res = torch.addcdiv(shift.to(f), m_new.to(f), denom.to(f), value=-lr * dc)   # addcdiv_stochastic_(shift, exp_avg, denom, value)
    shift1 = stochastic_round_bf16(res, draws[1])
    p_new = stochastic_round_bf16(shift1.to(f) + p.to(f), draws[2])      # add_stochastic_(p, shift)
    err = (p.to(f) - p_new.to(f)).to(bf)                                 # buffer.sub_(p)  (bf16)
    shift2 = stochastic_round_bf16(err.to(f) + shift1.to(f), draws[3])   # add_stochastic_(shift, buffer - p)
    if decay_this_iteration > 0:                                          # shift.add_(p, alpha=-decay)  (plain bf16 add)
        a32 = torch.tensor(-decay_this_iteration, dtype=f)
        if decay_alpha == "aten_cpu":
            abf = a32.to(bf).to(f)                                        # the CPU kernel holds alpha as a bf16 scalar
            n = shift2.numel()
            body = (shift2.to(f) + abf * p_new.to(f)).to(bf).flatten()   # 32-wide vector body: fp32 arithmetic on the widened lanes
            t0 = n - n % 32                                              # scalar tail: c10::BFloat16 arithmetic, the product is rounded too
            body[t0:] = (shift2.flatten()[t0:].to(f) + (abf * p_new.flatten()[t0:].to(f)).to(bf).to(f)).to(bf)
            shift2 = body.view(shift2.shape)
        else:
            shift2 = (shift2.to(f) + a32 * p_new.to(f)).to(bf)
    return p_new, m_new, v_new, shift2


# ------------------------------------------------------------------------------------------------
# loss variants and SNR helpers
# ------------------------------------------------------------------------------------------------
def conditional_loss_elementwise(pred, target, loss_type: str = "l2", huber_c=0.1):
    """ModelFoundation.conditional_loss with reduction='none' (common.py:6132-6166); huber_c may be a per-sample tensor [B] (scheduled)"""
    d2 = (pred.float() - target.float()) ** 2
    if loss_type == "l2":
        return d2
    c = huber_c if not torch.is_tensor(huber_c) else huber_c.float().view(-1, *([1] * (pred.dim() - 1)))
    root = torch.sqrt(d2 + c ** 2) - c
    if loss_type == "huber":
        return 2 * c * root
    if loss_type == "smooth_l1":
        return 2 * root
    raise NotImplementedError(loss_type)


def conditional_loss(pred, target, loss_type: str = "l2", huber_c=0.1, weight=None):
    """element loss -> [x per-sample weight] -> mean over CHW -> mean over the batch (common.py:6397-6398, 6426-6429)"""
    el = conditional_loss_elementwise(pred, target, loss_type, huber_c)
    if weight is not None:
        el = el * weight.float().view(-1, *([1] * (pred.dim() - 1)))
    per = el.mean(dim=list(range(1, el.dim())))
    return per.mean(), per


def scheduled_huber_c(timesteps, schedule: str, base_c: float, flow_matching: bool, num_train_timesteps: int = 1000, alphas_cumprod=None):
    """ModelFoundation.compute_scheduled_huber_c (common.py:6168-6216)"""
    if schedule == "constant":
        return torch.full_like(timesteps.float(), base_c)
    if schedule == "exponential":
        alpha = -math.log(base_c) / num_train_timesteps
        return torch.exp(-alpha * timesteps.float())
    if schedule == "snr":
        if flow_matching:
            s = timesteps.float() / 1000
            s = ((1.0 - s) / (s + 0.0001)) ** 0.5
        else:
            a = alphas_cumprod[timesteps.long()]
            s = ((1.0 - a) / a) ** 0.5
        return (1 - base_c) / (1 + s) ** 2 + base_c
    raise NotImplementedError(schedule)


def compute_snr(timesteps, alphas_cumprod, use_soft_min: bool = False, sigma_data: float = 1.0):
    """min_snr_gamma.py:4-41: (alpha/sigma)^2 with alpha = sqrt(acp[t]), sigma = sqrt(1 - acp[t]); soft-min variant"""
    alpha = (alphas_cumprod ** 0.5)[timesteps.long()].float()
    sigma = ((1.0 - alphas_cumprod) ** 0.5)[timesteps.long()].float()
    if use_soft_min:
        return (sigma * sigma_data) ** 2 / (sigma ** 2 + sigma_data ** 2) ** 2
    return (alpha / sigma) ** 2


# ------------------------------------------------------------------------------------------------
# fp8-native Linear (quantisation/fp8_native.py:25-119)
# ------------------------------------------------------------------------------------------------
FP8_E4M3_MAX = 448.0
FP8_E5M2_MAX = 57344.0


def fp8_quantize_weight(weight):
    """quantize_weight_to_fp8 (:25-30): per-output-row scale = max(amax, 1e-12) / 448; q = clamp(w / scale, +-448) -> e4m3fn"""
    w = weight.detach().float()
    amax = w.abs().amax(dim=1, keepdim=True).clamp(min=1e-12)
    scale = amax / FP8_E4M3_MAX
    q = (w / scale).clamp(-FP8_E4M3_MAX, FP8_E4M3_MAX).to(torch.float8_e4m3fn)
    return q, scale.squeeze(1)


def fp8_quantize_act(x_2d):
    """_Fp8NativeLinearFn.forward (:56-60).  The arithmetic runs on bf16 TENSORS in the reference, so the scale is a bf16 scalar and the
    scaled activations are rounded to bf16 before the e5m2 conversion; scale_a = float(bf16(1 / input_scale))."""
    input_scale = (FP8_E5M2_MAX / x_2d.detach().abs().amax().clamp(min=1e-12)).clamp(max=FP8_E5M2_MAX)
    x_q = (x_2d * input_scale).clamp(-FP8_E5M2_MAX, FP8_E5M2_MAX).to(torch.float8_e5m2)
    return x_q, input_scale.reciprocal().to(torch.float32)


def fp8_linear(x_q, scale_a, w_q, w_scale, bias=None, out_dtype=torch.bfloat16):
    """torch._scaled_mm(x_q, w_q.T, scale_a [M,1], scale_b [1,N], bias) with row-wise scales, accumulated in fp32 (:64-75)"""
    out = (x_q.float() @ w_q.float().t()) * scale_a.float().reshape(-1, 1) * w_scale.float().reshape(1, -1)
    if bias is not None:
        out = out + bias.float()
    return out.to(out_dtype)
