"""Shared checker code for the GPU parity tests and __graft_entry__.smoke(): builds a small Flux on the device, mirrors its
weights into the CPU oracle and compares.  (Test infrastructure: the only place besides bench.py's cpu_baseline leg that
touches oracle/.)"""
from __future__ import annotations

import torch

from oracle import flux as OF


def small_flux_cfg(layers=1, single=1, heads=2, head_dim=128, joint_dim=128, pooled=64, guidance=True):
    return dict(num_layers=layers, num_single_layers=single, num_attention_heads=heads, attention_head_dim=head_dim,
                joint_attention_dim=joint_dim, pooled_projection_dim=pooled, guidance_embeds=guidance, in_channels=64)


def oracle_state(model, dtype=torch.float32, device="cpu"):
    """device model -> ({name: tensor on `device`}, {target: (A, B)}, lora_scale).  device="cuda:0" runs the plain-torch fp32 oracle on the GPU's
    ATen kernels (full-width shapes, where the host cores would need minutes): still the restatement, not the product path"""
    P, A, B = {}, {}, {}
    for k, v in model.named_parameters():
        t = v.detach().to(device, dtype)
        if ".lora_A." in k:
            A[k.split(".lora_A.")[0]] = t
        elif ".lora_B." in k:
            B[k.split(".lora_B.")[0]] = t
        else:
            P[k] = t
    lora = {k: (A[k], B[k]) for k in A}
    scale = model.lora_groups[0].scale if model.lora_groups else 1.0
    return P, lora, scale


def oracle_cfg(model):
    c = model.config
    return OF.FluxConfig(in_channels=c.in_channels, num_layers=c.num_layers, num_single_layers=c.num_single_layers,
                         attention_head_dim=c.attention_head_dim, num_attention_heads=c.num_attention_heads,
                         joint_attention_dim=c.joint_attention_dim, pooled_projection_dim=c.pooled_projection_dim,
                         guidance_embeds=c.guidance_embeds, axes_dims_rope=c.axes_dims_rope)


def rel_l2(a, ref):
    a = a.detach().float().cpu(); ref = ref.detach().float().cpu()
    return ((a - ref).norm() / (ref.norm() + 1e-30)).item()


def cos_sim(a, ref):
    a = a.detach().float().cpu().flatten(); ref = ref.detach().float().cpu().flatten()
    return (torch.dot(a, ref) / (a.norm() * ref.norm() + 1e-30)).item()


def make_inputs(B, lat_h, lat_w, S_txt, joint_dim, pooled, device, seed=0, channels=16):
    g = torch.Generator().manual_seed(seed)
    latents = torch.randn(B, channels, lat_h, lat_w, generator=g)
    noise = torch.randn(B, channels, lat_h, lat_w, generator=g)
    prompt = torch.randn(B, S_txt, joint_dim, generator=g)
    pooled_t = torch.randn(B, pooled, generator=g)
    sigmas = torch.rand(B, generator=g) * 0.8 + 0.1
    bf = lambda t: t.to(torch.bfloat16)
    cpu = dict(latents=bf(latents).float(), noise=bf(noise).float(), prompt=bf(prompt).float(), pooled=bf(pooled_t).float(), sigmas=sigmas)
    dev = {k: (bf(v).to(device) if k != "sigmas" else v.to(device)) for k, v in dict(latents=latents, noise=noise, prompt=prompt, pooled=pooled_t, sigmas=sigmas).items()}
    return cpu, dev


def oracle_step(P, ocfg, lora, lora_scale, cpu, guidance_value=1.0, dtype=torch.float32, checkpoint=False):
    """reference step (on the device the weights P live on): noising -> model_predict -> MSE -> autograd.  Returns loss, prediction, {name: (dA, dB)}."""
    odev = next(iter(P.values())).device
    cpu = {k: v.to(odev) for k, v in cpu.items()}
    s = cpu["sigmas"].view(-1, 1, 1, 1).to(dtype)
    x, n = cpu["latents"].to(dtype), cpu["noise"].to(dtype)
    noisy = ((1 - s) * x + s * n).to(torch.bfloat16).to(dtype)       # the trainer feeds bf16 noisy latents (common.py:4990)
    target = (n - x).to(torch.bfloat16).to(dtype)
    Pd = P if all(v.dtype == dtype for v in P.values()) else {k: v.to(dtype) for k, v in P.items()}
    lp = {k: (a.to(dtype).requires_grad_(True), b.to(dtype).requires_grad_(True)) for k, (a, b) in lora.items()}
    pred = OF.flux_model_predict(Pd, ocfg, noisy, cpu["prompt"].to(dtype), cpu["pooled"].to(dtype), cpu["sigmas"].to(dtype) * 1000.0,
                                 guidance_value, lora=lp, lora_scale=lora_scale, checkpoint=checkpoint)
    loss = ((pred.float() - target.float()) ** 2).mean(dim=(1, 2, 3)).mean()
    loss.backward()
    grads = {k: (a.grad, b.grad) for k, (a, b) in lp.items()}
    return loss.detach(), pred.detach(), grads
