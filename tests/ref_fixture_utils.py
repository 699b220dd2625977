"""Helpers shared by tools/gen_ref_models.py (fixture generation from EXECUTED reference code) and the tests that replay the fixtures.

The "hip" tier of the fixtures does not store weights: both sides rebuild them from a seed with `seeded_state`, and the fixture carries a
checksum so a drift of the procedure (or of torch's CPU generator) fails loudly instead of silently comparing different networks."""
from __future__ import annotations

import torch


def seeded_state(named_shapes, seed: int):
    """name -> seeded tensor, drawn in sorted-name order from ONE CPU generator: weights N(0, 1/fan_in), biases N(0, 0.05^2), per-head q/k norm
    weights 1 + N(0, 0.1^2), PixArt scale_shift_tables N(0, 1/D).  No zero-initialised gate stays zero: every branch carries signal and gradient."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(named_shapes):
        shape = tuple(named_shapes[name])
        if "norm_q" in name or "norm_k" in name or "norm_added" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("scale_shift_table"):
            t = torch.randn(shape, generator=g) / shape[-1] ** 0.5
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / max(fan_in, 1) ** 0.5
        out[name] = t
    return out


def state_checksum(state) -> float:
    """order-independent fp64 checksum of a name -> tensor dict"""
    tot = 0.0
    for name in sorted(state):
        t = state[name].detach().double()
        tot += float((t * torch.arange(1, t.numel() + 1, dtype=torch.float64).reshape(t.shape).remainder(97.0)).sum())
    return tot


def seeded_lora(targets, shapes, rank: int, seed: int):
    """target -> (A [r, in], B [out, r]) for peft-style adapters: A ~ U(-1/sqrt(in), 1/sqrt(in)), B ~ N(0, 0.05^2) (nonzero so the delta is visible)"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in targets:
        o, i = shapes[name + ".weight"]
        A = (torch.rand(rank, i, generator=g) * 2 - 1) / i ** 0.5
        B = torch.randn(o, rank, generator=g) * 0.05
        out[name] = (A, B)
    return out


def rel_l2(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    return float((a - ref).norm() / (ref.norm() + 1e-300))
