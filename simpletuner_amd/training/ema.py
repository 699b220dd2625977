"""EMAModel — same constructor / step / copy_to / store / restore / state_dict surface as the reference's
simpletuner/helpers/training/ema.py:40-648, with the update executed by st355_ema_update (one launch over the flat
arena when the tracked parameters are contiguous, else one launch per tensor).

Decay schedule restated from ema.py:322-349; update `s -= (1-d)(s-p)` from ema.py:393-433 (K17).
Deviation (documented): the reference keeps the shadow on rank 0 only under DDP; replicas here are
bit-identical after the gradient all-reduce, so every rank may keep a shadow — `rank0_only=True` restores the reference
behaviour (other ranks skip the update).
"""
from __future__ import annotations

import copy
import weakref
from typing import Any, Dict, Iterable, Optional, Union

import torch

from .. import ops
from .optimizer import _contiguous_run


def should_update_ema(args, step) -> bool:
    """ema.py:29-37"""
    interval = getattr(args, "ema_update_interval", None)
    if interval is None:
        return True
    return step % interval == 0


class EMAModel:
    def __init__(self, args, accelerator, parameters: Iterable[torch.nn.Parameter], decay: float = 0.9999, min_decay: float = 0.0,
                 update_after_step: int = 0, warmup_steps: int = 0, use_ema_warmup: bool = False, inv_gamma: Union[float, int] = 1.0,
                 power: Union[float, int] = 2 / 3, foreach: bool = True, model_cls: Optional[Any] = None,
                 model_config: Dict[str, Any] = None, rank0_only: bool = False, **kwargs):
        parameters = list(parameters)
        self.args, self.accelerator = args, accelerator
        self.decay, self.min_decay = decay, min_decay
        self.update_after_step, self.warmup_steps = update_after_step, max(0, int(warmup_steps))
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0
        self.cur_decay_value = None
        self.temp_stored_params = None
        self.model_cls, self.model_config = model_cls, model_config
        self.rank0_only = rank0_only
        tracked = [p for p in parameters]
        self._tracked_param_ids = [id(p) for p in tracked]
        self._tracked_refs = [weakref.ref(p) for p in tracked]      # an id() is only meaningful while its object lives: a freed parameter's id gets reused
        self._flat = _contiguous_run([p.data for p in tracked]) and len(tracked) > 0
        if self._flat:
            n = sum(p.numel() for p in tracked)
            self.shadow_flat = torch.as_strided(tracked[0].data, (n,), (1,)).clone()
            self.shadow_params, off = [], 0
            for p in tracked:
                self.shadow_params.append(self.shadow_flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        else:
            self.shadow_flat = None
            self.shadow_params = [p.clone().detach() for p in tracked]

    # ---- ema.py:322-349 ----
    def get_decay(self, optimization_step: int = None) -> float:
        if optimization_step is None:
            optimization_step = self.optimization_step
        step = max(0, optimization_step - self.update_after_step - 1)
        if self.warmup_steps > 0:
            if optimization_step < self.warmup_steps:
                return 0.0
            return self.decay
        if step <= 0:
            return 0.0
        if self.use_ema_warmup:
            cur = 1 - (1 + step / self.inv_gamma) ** -self.power
        else:
            cur = (1 + step) / (10 + step)
        cur = min(cur, self.decay)
        return max(cur, self.min_decay)

    # ---- ema.py:352-433 ----
    @torch.no_grad()
    def step(self, parameters: Iterable[torch.nn.Parameter], global_step: int = None):
        if not should_update_ema(self.args, global_step):
            return
        if self.rank0_only and getattr(self.accelerator, "process_index", 0) != 0:
            return
        parameters = list(parameters)
        if len(parameters) != len(self.shadow_params):
            raise ValueError("EMAModel.step: parameter list does not match the tracked shadow parameters")
        if global_step is not None:
            self.optimization_step = global_step
        else:
            self.optimization_step += 1
        decay = self.get_decay(self.optimization_step)
        self.cur_decay_value = decay
        if self._flat and _contiguous_run([p.data for p in parameters]):
            n = self.shadow_flat.numel()
            ops.ema_update(self.shadow_flat, torch.as_strided(parameters[0].data, (n,), (1,)), decay)
            return
        for s, p in zip(self.shadow_params, parameters):
            if p.requires_grad:
                ops.ema_update(s.view(-1), p.data.to(s.dtype).contiguous().view(-1), decay)
            else:
                s.copy_(p.data.to(s.dtype))

    # ---- fused form: the update rides in the optimizer's own launch (st355_adamw_ema_step*: `ema` pointer + decay) ----
    def fused_decay(self, parameters, global_step: int):
        """The decay `step(parameters, global_step)` would apply, or None when that call would not be ONE flat-arena update of these parameters (interval skip,
        rank0_only on another rank, non-contiguous parameters).  Pure: the trainer hands (shadow_flat, decay) to the optimizer — decay is known before
        `optimizer.step()` because it depends on the step count alone (ema.py:322-349) — and calls `commit_fused` once the fused launch has run."""
        if not should_update_ema(self.args, global_step):
            return None
        if self.rank0_only and getattr(self.accelerator, "process_index", 0) != 0:
            return None
        parameters = list(parameters)
        if len(parameters) != len(self.shadow_params) or not self._flat or not _contiguous_run([p.data for p in parameters]):
            return None
        if parameters[0].dtype != self.shadow_flat.dtype:
            return None
        return self.get_decay(global_step)

    def commit_fused(self, global_step: int, decay: float) -> None:
        """bookkeeping of `step()` after the optimizer's launch applied s -= (1 - decay)(s - p_new) itself"""
        self.optimization_step = global_step
        self.cur_decay_value = decay

    def _align(self, parameters, allow_subset: bool):
        """ema.py:189-234: shadows are matched to the caller's parameters by IDENTITY, so reordered lists, subsets (allow_subset) and supersets
        with untracked entries all land on the right tensors.  Deviation, on purpose: when NONE of the given tensors is a tracked parameter but
        the count matches (a freshly built copy of the model), the match is positional — the reference silently copies nothing in that case."""
        params = list(parameters)
        by_id = {id(r()): s for r, s in zip(self._tracked_refs, self.shadow_params) if r() is not None}
        hits = [(by_id[id(p)], p) for p in params if id(p) in by_id]
        if not hits and len(params) == len(self.shadow_params):
            return list(zip(self.shadow_params, params))
        if not allow_subset:
            if len(params) != len(self.shadow_params):
                raise RuntimeError(f"EMA parameter count mismatch: expected {len(self.shadow_params)} parameters but received {len(params)}.")
            if len(hits) != len(params):
                raise RuntimeError(f"EMA parameter mapping failed: received {len(params) - len(hits)} untracked parameter(s). "
                                   "This usually means the model parameters were recreated after EMA initialization.")
        return hits

    def copy_to(self, parameters: Iterable[torch.nn.Parameter]) -> None:
        for s, p in self._align(parameters, allow_subset=True):
            p.data.copy_(s.to(device=p.device, dtype=p.dtype))

    def store(self, parameters: Iterable[torch.nn.Parameter]) -> None:
        parameters = list(parameters)
        self.temp_stored_params = [p.detach().clone() for p in parameters]
        self._temp_stored_param_ids = [weakref.ref(p) for p in parameters]

    def restore(self, parameters: Iterable[torch.nn.Parameter]) -> None:
        """ema.py:540-609: stored copies go back to the SAME parameter objects, whatever order they are passed in"""
        if self.temp_stored_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        parameters = list(parameters)
        by_id = {id(r()): c for r, c in zip(self._temp_stored_param_ids, self.temp_stored_params) if r() is not None}
        if all(id(p) in by_id for p in parameters):
            pairs = [(by_id[id(p)], p) for p in parameters]
        elif not any(id(p) in by_id for p in parameters) and len(parameters) == len(self.temp_stored_params):
            pairs = list(zip(self.temp_stored_params, parameters))
        else:
            missing = sum(id(p) not in by_id for p in parameters)
            raise RuntimeError(f"EMA restore failed: received {missing} untracked parameter(s). "
                               "This usually means the model parameters were recreated after EMA.store().")
        for c, p in pairs:
            p.data.copy_(c.data)
        self.temp_stored_params = None
        self._temp_stored_param_ids = None

    def to(self, device=None, dtype=None, non_blocking=False):
        return self   # shadows live next to the parameters in HBM (288 GB: no CPU shuttle, ema.py:357-359/432-433 not needed)

    def pin_memory(self) -> None:
        """ema.py:474-492 pins the shadow for host offload; the shadow lives in HBM on this path: nothing to pin"""
        return None

    def save_pretrained(self, path, max_shard_size: str = "10GB"):
        """ema.py:298-320 writes a diffusers model folder from `model_cls` / `model_config`; this path saves the shadow in the reference's
        `ema_model.pt` dict layout next to the weights (Trainer.save_state) — the folder form needs the diffusers class and is refused loudly"""
        raise NotImplementedError("EMAModel.save_pretrained: use save_state_dict(<dir>/ema_model.pt) (ema.py:236-296 layout)")

    def parameter_count(self) -> int:
        return sum(p.numel() for p in self.shadow_params)

    def state_dict(self, destination=None, prefix: str = "", keep_vars: bool = False, exclude_params: bool = False) -> dict:
        """ema.py:500-524 layout: scalars + `shadow_params.{i}`"""
        sd = dict(decay=self.decay, min_decay=self.min_decay, optimization_step=self.optimization_step,
                  update_after_step=self.update_after_step, warmup_steps=self.warmup_steps, use_ema_warmup=self.use_ema_warmup,
                  inv_gamma=self.inv_gamma, power=self.power)
        if exclude_params:
            return sd
        for i, s in enumerate(self.shadow_params):
            sd[f"{prefix}shadow_params.{i}"] = s if keep_vars else s.detach().clone()
        return sd

    def save_state_dict(self, path: str) -> None:
        """ema.py:236-249: the `ema_model.pt` of a checkpoint (save_hooks.py:396-418) — torch.save of state_dict(), tensors on the host so the file
        loads anywhere"""
        import os
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        torch.save({k: (v.detach().to("cpu") if torch.is_tensor(v) else v) for k, v in self.state_dict().items()}, path)

    def load_state_dict(self, state) -> None:
        """ema.py:251-286: `state` is the path of an `ema_model.pt` (the reference's call, save_hooks.py:443) or an already loaded dict; the number
        of shadow tensors must match"""
        sd = torch.load(state, map_location="cpu", weights_only=True) if isinstance(state, (str, bytes)) or hasattr(state, "__fspath__") else copy.copy(state)
        for k in ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma", "power"):
            if k in sd:
                setattr(self, k, sd[k])
        if "warmup_steps" in sd:
            self.warmup_steps = max(0, int(sd["warmup_steps"] or 0))
        n = 0
        while f"shadow_params.{n}" in sd:
            n += 1
        if n != len(self.shadow_params):
            raise ValueError(f"Mismatch in number of shadow parameters: expected {len(self.shadow_params)}, but found {n} in the state dict.")
        with torch.no_grad():
            for i, s in enumerate(self.shadow_params):
                s.copy_(sd[f"shadow_params.{i}"].to(device=s.device, dtype=s.dtype))

    # nn.Module-ish iterators the reference's hooks rely on (ema.py:611-648)
    def parameters(self, recurse: bool = True):
        return iter(self.shadow_params)

    def named_parameters(self, prefix: str = "", recurse: bool = True):
        for i, s in enumerate(self.shadow_params):
            yield f"{prefix}shadow_params.{i}", s

    def buffers(self, recurse: bool = True):
        return iter([])

    def named_buffers(self, prefix: str = "", recurse: bool = True):
        return iter([])

    def children(self):
        return iter([])

    def named_children(self):
        return iter([])

    def modules(self):
        yield self

    def named_modules(self, memo=None, prefix: str = ""):
        yield prefix, self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def zero_grad(self):
        pass
