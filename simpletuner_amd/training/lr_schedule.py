"""Learning-rate schedules of the step loop (`lr_scheduler.step()` after every optimizer step, trainer.py:7239-7260): the reference's own
classes in simpletuner/helpers/training/custom_schedule.py — `get_polynomial_decay_schedule_with_warmup` :102-154, `Cosine` :195-280,
`CosineAnnealingHardRestarts` :283-386, `Sine` :389-440 — selected by `get_lr_scheduler` :481-557, restated as closed forms of the step number.

Why closed forms: on this path the learning rate is a scalar kernel argument of the one fused optimizer launch, so a schedule is just
`lr(step)` on the host; no per-group tensor state, trivially resumable (`state_dict` is the step counter).  The reference classes'
observable quirks are kept, because they define the curves users get:
  * `sine`:   eta_min + (base - eta_min) * (1 + sin(pi * step / T_0)) / 2   — starts at the MIDPOINT, first peak at T_0/2, period 2*T_0;
  * `cosine`: with the default `steps_per_epoch = -1` the class computes T_cur = -step, and cos is even: eta_min + (base - eta_min) *
              (1 + cos(pi * step / T_0)) / 2 — an un-restarted cosine of period 2*T_0 (it never reaches the restart branch);
  * `cosine_with_restarts`: `step % -1 == 0` for every step, so T_cur is always 0 and the rate stays at base (the reference warns that this
              scheduler "is currently misbehaving"; reproduced, not fixed);
  * all three truncate the rate to 1e-9 (`floor(lr * 1e9) / 1e9`);
  * `polynomial`: linear warm-up, then lr_end + (base - lr_end) * (1 - progress)^power, lr_end after the last step; warm-up and total are
              multiplied by the process count by the caller, as in the reference.
`constant` / `constant_with_warmup` / `linear` come from diffusers' `get_scheduler` in the reference (un-vendored); their textbook forms are
restated and marked unpinned.  Pinned: tests/golden/lr_schedule_vectors.pt = the reference classes executed by tools/gen_golden.py.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional


def _trunc9(x: float) -> float:
    return math.floor(x * 1e9) / 1e9


def sine_lr(step: int, base_lr: float, T_0: int, eta_min: float = 0.0) -> float:
    return _trunc9(eta_min + (base_lr - eta_min) * (0.5 * (1 + math.sin(math.pi * step / T_0))))


def cosine_lr(step: int, base_lr: float, T_0: int, eta_min: float = 0.0) -> float:
    t_cur = (step // -1) + (step % -1) / -1                      # the class's arithmetic with steps_per_epoch = -1: exactly -step
    return _trunc9(eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / T_0)) / 2)


def cosine_hard_restarts_lr(step: int, base_lr: float, T_0: int, eta_min: float = 0.0) -> float:
    t_cur = step % -1                                             # always 0: see the module docstring
    return _trunc9(eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / T_0)) / 2)


def polynomial_factor(step: int, lr_init: float, num_warmup_steps: int, num_training_steps: int, lr_end: float = 1e-7, power: float = 1.0) -> float:
    """the LambdaLR multiplier of get_polynomial_decay_schedule_with_warmup (:143-154)"""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    if step > num_training_steps:
        return float(lr_end) / float(lr_init)
    span = int(num_training_steps) - int(num_warmup_steps)
    remaining = 1 - (step - int(num_warmup_steps)) / span
    return ((float(lr_init) - float(lr_end)) * remaining ** power + float(lr_end)) / float(lr_init)


class St355LRSchedule:
    """`torch.optim.lr_scheduler`-shaped driver around `lr_of(step, base_lr) -> lr`: writes `param_group["lr"]` (read by the fused optimizer
    launch as a scalar), `step()`, `get_last_lr()`, `state_dict()` / `load_state_dict()` (the counter and the base rates: resume is exact)."""

    def __init__(self, optimizer, lr_of: Callable[[int, float], float], last_step: int = -1, name: str = "custom"):
        self.optimizer, self.lr_of, self.name = optimizer, lr_of, name
        self.base_lrs: List[float] = [g.setdefault("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.last_epoch = last_step
        self._last_lr = [g["lr"] for g in optimizer.param_groups]
        self.step()                                               # like LRScheduler.__init__: the first step() sets the rate of step last+1

    def step(self, step: Optional[int] = None):
        self.last_epoch = self.last_epoch + 1 if step is None else int(step)
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = self.lr_of(self.last_epoch, base)
        self._last_lr = [g["lr"] for g in self.optimizer.param_groups]

    def get_last_lr(self) -> List[float]:
        return list(self._last_lr)

    def state_dict(self) -> Dict:
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs), "_last_lr": list(self._last_lr), "name": self.name}

    def load_state_dict(self, sd: Dict) -> None:
        self.last_epoch = int(sd["last_epoch"])
        self.base_lrs = list(sd.get("base_lrs", self.base_lrs))
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = self.lr_of(self.last_epoch, base)
        self._last_lr = [g["lr"] for g in self.optimizer.param_groups]


def get_lr_scheduler(args, optimizer, accelerator, logger=None, global_step: int = 0, use_deepspeed_scheduler: bool = False) -> St355LRSchedule:
    """custom_schedule.py:481-557, same argument list.  T_0 = lr_warmup_steps * num_processes for the periodic schedules (the reference
    reuses the warm-up setting as the half period)."""
    if use_deepspeed_scheduler:
        raise NotImplementedError("DeepSpeed schedulers are outside this path")
    world = int(getattr(accelerator, "num_processes", 1) or 1)
    name = getattr(args, "lr_scheduler", "constant")
    eta_min = float(getattr(args, "lr_end", 0.0) or 0.0)
    t0 = int(getattr(args, "lr_warmup_steps", 0) * world)
    if name in ("sine", "cosine_with_restarts"):
        if t0 <= 0:
            raise ValueError(f"{'Sine learning rate expects' if name == 'sine' else 'Expected'} positive integer T_0, but got {t0}")
        fn = sine_lr if name == "sine" else cosine_hard_restarts_lr
        return St355LRSchedule(optimizer, lambda s, base: fn(s, base, t0, eta_min), name=name)
    if name == "cosine":
        period = t0 if t0 > 0 else 1000                          # the class falls back to 1000 with a warning (:222-226)
        return St355LRSchedule(optimizer, lambda s, base: cosine_lr(s, base, period, eta_min), name=name)
    if name == "polynomial":
        lr_init = optimizer.defaults["lr"]
        lr_end = getattr(args, "lr_end", 1e-7)
        if not (float(lr_init) > float(lr_end)):
            raise ValueError(f"lr_end ({lr_end}) must be be smaller than initial lr ({lr_init})")
        warm, total, power = t0, int(args.max_train_steps * world), float(getattr(args, "lr_power", 1.0))
        return St355LRSchedule(optimizer, lambda s, base: base * polynomial_factor(s, lr_init, warm, total, lr_end, power), last_step=global_step - 1, name=name)
    # diffusers.optimization.get_scheduler names (un-vendored; textbook forms, unpinned)
    total = int(getattr(args, "max_train_steps", 0) * world)
    if name == "constant":
        return St355LRSchedule(optimizer, lambda s, base: base, name=name)
    if name == "constant_with_warmup":
        return St355LRSchedule(optimizer, lambda s, base: base * (min(1.0, s / max(1.0, t0)) if t0 > 0 else 1.0), name=name)
    if name == "linear":
        return St355LRSchedule(optimizer, lambda s, base: base * (s / max(1, t0) if s < t0 else max(0.0, (total - s) / max(1, total - t0))), name=name)
    raise NotImplementedError(f"lr_scheduler '{name}' is not implemented on the st355 path")
