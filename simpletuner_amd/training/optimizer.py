"""Fused AdamW(+EMA) optimizer — the entry a maintainer adds to `optimizer_choices`
(simpletuner/helpers/training/optimizer_param.py:76-96): {"precision": "any", "default_settings": {...}, "class": St355AdamW}.

torch.optim.AdamW semantics (decoupled decay, bias correction; optimizer_param.py:87-96 defaults), executed by
st355_adamw_ema_step: when the group's parameters are views of one contiguous arena (the LoRA flat arena, or a bf16
full-fine-tune arena) the whole step is ONE launch over the arena; otherwise one launch per tensor.  Standard
torch.optim.Optimizer API so `accelerator.prepare(optimizer)` / `accelerator.save_state` work unchanged.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops

F32 = torch.float32


def _contiguous_run(tensors):
    """True if `tensors` are back-to-back views of one allocation (same dtype), in order."""
    if not tensors:
        return False
    dt = tensors[0].dtype
    ptr = tensors[0].data_ptr()
    for t in tensors:
        if t.dtype != dt or not t.is_contiguous() or t.data_ptr() != ptr:
            return False
        ptr += t.numel() * t.element_size()
    return True


class St355AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 amsgrad: bool = False, **_ignored):
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented in the fused kernel")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0           # set by the gradient-sync layer (1/world_size) or by clipping: folded into the kernel
        self.ema_shadow_flat: Optional[torch.Tensor] = None   # optional fused EMA (flat arena path only)
        self.ema_decay = 0.0
        self._flat = {}

    def _group_flat(self, gi, group):
        st = self._flat.get(gi)
        if st is None:
            ps = [p for p in group["params"] if p.requires_grad]
            ok = _contiguous_run([p.data for p in ps])
            n = sum(p.numel() for p in ps)
            st = dict(ok=ok, ps=ps, n=n, step=0, m=None, v=None)
            if ok:
                st["m"] = torch.zeros(n, dtype=F32, device=ps[0].device)
                st["v"] = torch.zeros(n, dtype=F32, device=ps[0].device)
                off = 0
                for p in ps:   # torch-compatible per-parameter state = views of the flat moments
                    self.state[p] = dict(step=torch.tensor(0.0), exp_avg=st["m"][off:off + p.numel()].view_as(p),
                                         exp_avg_sq=st["v"][off:off + p.numel()].view_as(p))
                    off += p.numel()
            self._flat[gi] = st
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            st = self._group_flat(gi, group)
            ps = [p for p in st["ps"] if p.grad is not None]
            if not ps:
                continue
            b1, b2 = group["betas"]
            st["step"] += 1
            step = st["step"]
            grads = [p.grad for p in ps]
            if st["ok"] and len(ps) == len(st["ps"]) and _contiguous_run(grads):
                pflat = torch.as_strided(ps[0].data, (st["n"],), (1,))
                gflat = torch.as_strided(grads[0], (st["n"],), (1,))
                ops.adamw_ema_step(pflat, gflat, st["m"], st["v"], step, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                   grad_scale=self.grad_scale, ema=self.ema_shadow_flat, ema_decay=self.ema_decay)
                for p in ps:
                    self.state[p]["step"] += 1
                continue
            for p in ps:   # generic path: one launch per tensor
                s = self.state[p]
                if "exp_avg" not in s:
                    s["step"] = torch.tensor(0.0)
                    s["exp_avg"] = torch.zeros(p.numel(), dtype=F32, device=p.device).view_as(p)
                    s["exp_avg_sq"] = torch.zeros(p.numel(), dtype=F32, device=p.device).view_as(p)
                s["step"] += 1
                g = p.grad.contiguous()
                if g.dtype != p.dtype:
                    g = g.to(p.dtype)
                ops.adamw_ema_step(p.data.view(-1), g.view(-1), s["exp_avg"].view(-1), s["exp_avg_sq"].view(-1), int(s["step"].item()),
                                   group["lr"], b1, b2, group["eps"], group["weight_decay"], grad_scale=self.grad_scale)
        return loss


# what `optimizer_choices["st355-adamw"]` looks like in the reference's registry (optimizer_param.py:76-96)
OPTIMIZER_CHOICE = {
    "st355-adamw": {
        "precision": "any",
        "default_settings": {"betas": (0.9, 0.999), "weight_decay": 1e-2, "eps": 1e-8},
        "class": St355AdamW,
    }
}
