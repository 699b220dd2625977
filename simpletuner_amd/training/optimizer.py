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


def _unpack_saved_state(opt: torch.optim.Optimizer, state_dict: dict):
    """torch.optim.Optimizer.load_state_dict's matching rules (groups by position, parameters by position inside a group; saved hyper-parameters
    replace the live ones) WITHOUT its dtype policy: torch casts every floating state tensor to the parameter's dtype, which would round the fp32
    moments of bf16 parameters to bf16 on every resume.  Returns {param: saved per-parameter state}."""
    saved_groups = state_dict["param_groups"]
    if len(saved_groups) != len(opt.param_groups):
        raise ValueError("loaded state dict has a different number of parameter groups")
    by_param = {}
    for saved, live in zip(saved_groups, opt.param_groups):
        if len(saved["params"]) != len(live["params"]):
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
        for k, v in saved.items():
            if k != "params":
                live[k] = v
        for pid, p in zip(saved["params"], live["params"]):
            if pid in state_dict["state"]:
                by_param[p] = state_dict["state"][pid]
    return by_param


class St355AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 amsgrad: bool = False, **_ignored):
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented in the fused kernel")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0           # set by the gradient-sync layer (1/world_size) or by clipping: folded into the kernel
        self.ema_shadow_flat: Optional[torch.Tensor] = None   # optional fused EMA (flat arena path only)
        self.ema_decay = 0.0
        self.ema_applied = False
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
    def load_state_dict(self, state_dict: dict) -> None:
        """resume (`accelerator.load_state`, save_hooks.py): the saved per-parameter moments are copied INTO the flat fp32 arenas (created here if
        the optimizer has not stepped yet) so the one-launch path continues from them; step counters are restored"""
        saved = _unpack_saved_state(self, state_dict)
        self._flat = {}
        for p in list(self.state):
            del self.state[p]
        for gi, group in enumerate(self.param_groups):
            st = self._group_flat(gi, group)
            steps = [0]
            for p in st["ps"]:
                old = saved.get(p)
                if old is None:
                    continue
                k = int(float(old["step"]))
                steps.append(k)
                if st["ok"]:
                    mine = self.state[p]
                    mine["exp_avg"].copy_(old["exp_avg"].to(device=p.device, dtype=F32).view_as(p))
                    mine["exp_avg_sq"].copy_(old["exp_avg_sq"].to(device=p.device, dtype=F32).view_as(p))
                    mine["step"] = torch.tensor(float(k))
                else:
                    self.state[p] = dict(step=torch.tensor(float(k)), exp_avg=old["exp_avg"].to(device=p.device, dtype=F32).clone().view_as(p),
                                         exp_avg_sq=old["exp_avg_sq"].to(device=p.device, dtype=F32).clone().view_as(p))
            st["step"] = max(steps)

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
                ema = self.ema_shadow_flat if (self.ema_shadow_flat is not None and len(self.param_groups) == 1
                                               and self.ema_shadow_flat.numel() == st["n"] and self.ema_shadow_flat.dtype == pflat.dtype) else None
                ops.adamw_ema_step(pflat, gflat, st["m"], st["v"], step, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                   grad_scale=self.grad_scale, ema=ema, ema_decay=self.ema_decay)
                self.ema_applied = ema is not None      # the trainer falls back to EMAModel.step when the fused form did not run
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


class St355AdamWBF16(torch.optim.Optimizer):
    """AdamWBF16 — the reference examples' default optimizer (optimizers/adamw_bfloat16/__init__.py:20-111), as ONE fused launch.

    Same constructor (keyword-only lr, betas, eps, weight_decay), same `step(zero_grad=False)`, same per-parameter state keys
    (`step`, `exp_avg`, `exp_avg_sq`, `shift`, `accumulated_decay`) so `accelerator.save_state` round-trips; the states are views of
    flat bf16 arenas.  The per-tensor delayed-decay schedule (decay owed += weight_decay*lr; applied only above 5e-3; random initial
    phase per tensor) is host arithmetic exactly as in the reference; the element-wise math is st355_adamw_bf16_sr_step.
    Parameters must be bf16 and, for the fused path, views of one contiguous arena (as the full-fine-tune engine allocates them);
    otherwise one launch per tensor.  `rand_bits_hook(p_index, step) -> int32[4, n]` lets parity tests inject the reference's draws."""
    decay_threshold = 5e-3

    def __init__(self, params, *, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, seed: int = 0):
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        super().__init__(params, dict(betas=betas, eps=eps, weight_decay=weight_decay, lr=lr))
        self.grad_scale = 1.0
        self.seed = int(seed)
        self.rand_bits_hook = None
        self._launches = 0
        self._flat = {}

    def _init_group(self, gi, group):
        ps = [p for p in group["params"] if p.requires_grad]
        for p in ps:
            assert p.dtype == torch.bfloat16, "only bfloat 16 is supported."
        ok = _contiguous_run([p.data for p in ps])
        n = sum(p.numel() for p in ps)
        dev = ps[0].device
        st = dict(ok=ok, ps=ps, n=n, step=0)
        mk = lambda: torch.zeros(n, dtype=torch.bfloat16, device=dev)
        st["m"], st["v"], st["shift"] = mk(), mk(), mk()
        ends, off = [], 0
        # "Each weight has its own starting point to avoid simultaneous updates in all weights" (:80-84).  The reference draws these phases from the
        # global torch RNG; here they come from a generator private to (optimizer seed, group), so data-parallel replicas — whose global RNG streams
        # differ by rank — release their delayed decay on the same steps and stay bit-identical.
        phase_rng = torch.Generator().manual_seed(1_000_003 * self.seed + gi)
        for p in ps:
            k = p.numel()
            self.state[p] = dict(step=0.0, exp_avg=st["m"][off:off + k].view_as(p), exp_avg_sq=st["v"][off:off + k].view_as(p),
                                 shift=st["shift"][off:off + k].view_as(p),
                                 accumulated_decay=float(torch.rand([], generator=phase_rng) * self.decay_threshold))
            off += k
            ends.append(off)
        st["seg_end"] = torch.tensor(ends, dtype=torch.int64, device=dev)
        self._flat[gi] = st
        return st

    @torch.no_grad()
    def load_state_dict(self, state_dict: dict) -> None:
        """resume: moments / shift copied into the flat bf16 arenas, per-tensor step and the owed decay restored (the delayed-decay phase of every
        tensor continues where it stopped, optimizers/adamw_bfloat16/__init__.py:80-95)"""
        saved = _unpack_saved_state(self, state_dict)
        self._flat = {}
        for p in list(self.state):
            del self.state[p]
        for gi, group in enumerate(self.param_groups):
            st = self._init_group(gi, group)
            steps = [0]
            for p in st["ps"]:
                old = saved.get(p)
                if old is None:
                    continue
                mine = self.state[p]
                for k in ("exp_avg", "exp_avg_sq", "shift"):
                    mine[k].copy_(old[k].to(device=p.device, dtype=torch.bfloat16).view_as(p))
                mine["step"] = float(old["step"])
                mine["accumulated_decay"] = float(old["accumulated_decay"])
                steps.append(int(float(old["step"])))
            st["step"] = max(steps)

    @torch.no_grad()
    def step(self, zero_grad: bool = False, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi) or self._init_group(gi, group)
            ps = st["ps"]
            if any(p.grad is None for p in ps):
                raise RuntimeError("St355AdamWBF16 expects a gradient for every parameter of the group (fused arena step)")
            beta1, beta2 = group["betas"]
            lr = group["lr"]
            st["step"] += 1
            decays = []
            for p in ps:                                   # reference :89-95, host scalars
                s = self.state[p]
                s["step"] += 1
                s["accumulated_decay"] += group["weight_decay"] * lr
                acc = s["accumulated_decay"]
                dec = acc if acc > self.decay_threshold else 0.0
                s["accumulated_decay"] -= dec
                decays.append(dec)
            if any(d != 0.0 for d in decays):
                seg_decay = torch.tensor(decays, dtype=F32, device=ps[0].device)
            else:
                # no tensor releases its owed decay this step (the common case: weight_decay * lr per step against the 5e-3 threshold): a resident zero vector —
                # the host->device copy of the list is a blocking copy, i.e. a host sync in every step (6.8 ms of a 100 ms graph-replayed step, measured r5)
                seg_decay = st.get("zero_decay")
                if seg_decay is None:
                    seg_decay = st["zero_decay"] = torch.zeros(len(ps), dtype=F32, device=ps[0].device)
            grads = [p.grad for p in ps]
            fused = st["ok"] and _contiguous_run(grads) and self.rand_bits_hook is None
            if fused:
                pflat = torch.as_strided(ps[0].data, (st["n"],), (1,))
                gflat = torch.as_strided(grads[0], (st["n"],), (1,))
                ops.adamw_bf16_sr_step(pflat, gflat, st["m"], st["v"], st["shift"], st["step"], lr, beta1, beta2, group["eps"],
                                       seg_end=st["seg_end"], seg_decay=seg_decay, seed=self.seed, offset=4 * st["n"] * st["step"],
                                       grad_scale=self.grad_scale)
                self._launches += 1
            else:
                off = 0
                for i, p in enumerate(ps):
                    k = p.numel()
                    g = p.grad.contiguous().view(-1)
                    rb = self.rand_bits_hook(i, st["step"]) if self.rand_bits_hook is not None else None
                    ops.adamw_bf16_sr_step(p.data.view(-1), g, st["m"][off:off + k], st["v"][off:off + k], st["shift"][off:off + k],
                                           st["step"], lr, beta1, beta2, group["eps"], seg_end=st["seg_end"][i:i + 1] - off,
                                           seg_decay=seg_decay[i:i + 1], rand_bits=rb, seed=self.seed + i,
                                           offset=4 * k * st["step"], grad_scale=self.grad_scale)
                    self._launches += 1
                    off += k
            if zero_grad:
                for p in ps:
                    p.grad.zero_()
        return loss


# what `optimizer_choices["st355-adamw"]` looks like in the reference's registry (optimizer_param.py:76-96)
OPTIMIZER_CHOICE = {
    "st355-adamw": {
        "precision": "any",
        "default_settings": {"betas": (0.9, 0.999), "weight_decay": 1e-2, "eps": 1e-8},
        "class": St355AdamW,
    },
    # the reference's own "adamw_bf16" entry (optimizer_param.py), with the fused class substituted
    "adamw_bf16": {
        "precision": "bf16",
        "default_settings": {"betas": (0.9, 0.999), "weight_decay": 1e-2, "eps": 1e-6},
        "class": St355AdamWBF16,
    },
}
