"""The gradient-synchronisation seam under the reference's OWN Trainer.

The reference never builds a reducer itself: `Accelerator(kwargs_handlers=[DistributedDataParallelKwargs(...)])` (helpers/training/trainer.py:1034-1041),
`accelerator.prepare(model)` wraps the trained component in `torch.nn.parallel.DistributedDataParallel` (:4564-4571), hands the wrapper back through
`model.set_prepared_model` (:4577), runs each micro-step inside `accelerator.accumulate(model)` (:7009 — DDP's `no_sync()` on every micro-step but the
last) and reads `accelerator.sync_gradients` afterwards (:7139).  Two ways to keep that surface and still exchange gradients the MI355X way
(`training.grad_sync.GradSync`: the flat gradient arena reduced slice by slice on a dedicated comm stream WHILE the hand-written backward still runs):

  * `install_ddp_comm_hook(ddp)` — the simplest drop-in: the wrapper stays torch's DDP, exactly as accelerate built it, and gets a DDP communication
    hook.  On a step that follows no un-synchronised micro-step, the st355 backward has already summed AND averaged the arena over the ranks by the
    time DDP's reducer calls the hook; the hook then returns the bucket untouched (no second exchange, no extra pass).  On the boundary step of a
    gradient accumulation (locally accumulated `.grad` from `no_sync()` micro-steps sits in the buckets) the hook all-reduces and averages the bucket
    itself — DDP's default behaviour, i.e. the exact semantics of trainer.py:7009.
  * `St355DistributedDataParallel(module)` — the cleanest: a wrapper with the attributes accelerate and the reference touch (`.module`, `no_sync()`,
    `require_backward_grad_sync`, `forward`, `state_dict` keys under `module.`), no reducer, no bucket copies: the arena IS the bucket.  Boundary
    steps of an accumulation reduce the accumulated flat `.grad` once, after the backward (a callback queued on the autograd engine).

Both produce, after every synchronised backward, `param.grad == mean over ranks of the (accumulated) local gradients` — DDP's contract — so the reference's
`accelerator.clip_grad_norm_`, its optimizers and its logging see what they would see under DDP.  (The in-repo Trainer keeps SUMs and folds 1/world into
the optimizer kernel instead; that is `average=False`.)

hipGraph capture with N > 1: the collectives are issued on the comm stream behind events of the capture stream, so `torch.cuda.graph` (which captures
every stream that joins the capture through an event) records them as graph nodes — RCCL supports capture; `capture_safe()` says whether the active
backend does (nccl == RCCL: yes; gloo: no, its collectives run on host threads)."""
from __future__ import annotations

import contextlib
from typing import Optional

import torch
import torch.distributed as dist

from .grad_sync import GradSync, sync_module_states


def _arena_of(comp):
    """the flat gradient arena the component's backward fills (full fine-tune arena, else the LoRA arena)"""
    if getattr(comp, "full", False) and getattr(comp, "grad_arena", None) is not None:
        return comp.grad_arena, 128 << 20
    if getattr(comp, "lora_grad_flat", None) is not None:
        return comp.lora_grad_flat, 32 << 20
    if getattr(comp, "grad_arena", None) is not None:
        return comp.grad_arena, 128 << 20
    raise TypeError(f"{type(comp).__name__} exposes no flat gradient arena (lora_grad_flat / grad_arena): not an st355 trained component")


def attach_grad_sync(comp, process_group=None, bucket_bytes: Optional[int] = None, mode: str = "auto", comm=None, fp32_reduce: Optional[bool] = None) -> GradSync:
    arena, default_bucket = _arena_of(comp)
    gs = GradSync(arena, bucket_bytes=bucket_bytes or default_bucket, process_group=process_group, mode=mode, comm=comm, fp32_reduce=fp32_reduce)
    comp.grad_sync = gs
    return gs


def capture_safe(process_group=None) -> bool:
    """may the gradient exchange be captured into a hipGraph together with the backward?  RCCL: yes (stream-ordered, capturable); gloo: no."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return True
    return str(dist.get_backend(process_group)).lower() == "nccl"


class _SeamState:
    """shared by the wrapper and the comm hook: which micro-steps ran without synchronisation since the last synchronised backward"""

    def __init__(self, comp, average: bool):
        self.comp, self.average = comp, average
        self.pending_unsynced = False            # a no_sync() micro-step has accumulated local gradients that no exchange has seen yet
        self.reduced_in_backward = False         # this backward's arena was exchanged (and averaged) while it ran

    def before_forward(self, sync_this_step: bool):
        gs = self.comp.grad_sync
        overlapped = sync_this_step and not self.pending_unsynced
        gs.enabled = overlapped
        # 1/world is applied in the SAME pass that hands the arena to autograd (the private flat copy every backward makes anyway): no extra HBM pass
        self.comp._handover_scale = (1.0 / gs.world_size) if (overlapped and self.average) else None
        self.reduced_in_backward = overlapped
        if not sync_this_step:
            self.pending_unsynced = True


class St355DistributedDataParallel(torch.nn.Module):
    """`accelerator.prepare`'s return value for an st355 trained component without torch's reducer (module docstring)."""

    def __init__(self, module: torch.nn.Module, process_group=None, bucket_bytes: Optional[int] = None, mode: str = "auto", comm=None,
                 average: bool = True, fp32_reduce: Optional[bool] = None, **_ddp_kwargs_ignored):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.require_backward_grad_sync = True
        sync_module_states(module, 0, process_group)                       # DDP construction semantics: every replica starts from rank 0's state
        self.grad_sync = attach_grad_sync(module, process_group, bucket_bytes, mode, comm, fp32_reduce)
        self._seam = _SeamState(module, average)
        module._post_backward_cb = self._after_backward

    # the attributes accelerate / the reference read on a DDP-wrapped model
    @contextlib.contextmanager
    def no_sync(self):
        old = self.require_backward_grad_sync
        self.require_backward_grad_sync = False
        try:
            yield
        finally:
            self.require_backward_grad_sync = old

    def forward(self, *args, **kwargs):
        self._sync_this_backward = bool(self.require_backward_grad_sync) and torch.is_grad_enabled()
        if torch.is_grad_enabled():
            self._seam.before_forward(self._sync_this_backward)
        return self.module(*args, **kwargs)

    def _after_backward(self):
        """queued on the autograd engine by the component's backward: runs once the whole backward (incl. AccumulateGrad) has finished"""
        st = self._seam
        if not getattr(self, "_sync_this_backward", False):
            return
        if st.pending_unsynced and not st.reduced_in_backward:
            # boundary of a gradient accumulation: ONE exchange of the accumulated flat .grad (DDP semantics of trainer.py:7009)
            params = (self.module.trainable_parameters() if hasattr(self.module, "trainable_parameters")
                      else [p for p in self.module.parameters() if p.requires_grad])
            grads = [p.grad for p in params]
            from .optimizer import _contiguous_run
            if any(g is None for g in grads) or not _contiguous_run(grads):
                raise RuntimeError("accumulated gradients are not one flat arena (st355 components hand autograd one flat buffer per backward)")
            flat = torch.as_strided(grads[0], (sum(g.numel() for g in grads),), (1,))
            W = self.grad_sync.world_size
            if W > 1:
                self.grad_sync.all_reduce_now(flat)
                if st.average:
                    flat.mul_(1.0 / W)
        st.pending_unsynced = False


class _HookState:
    def __init__(self, seam: _SeamState, process_group):
        self.seam, self.pg = seam, process_group


def _comm_hook(state: _HookState, bucket):
    """torch DDP communication hook (`register_comm_hook` contract: returns a Future of the bucket's reduced flat tensor)"""
    seam = state.seam
    buf = bucket.buffer()
    if seam.reduced_in_backward:
        # the st355 backward exchanged and averaged this gradient while it ran (GradSync on the comm stream): nothing left to do for this bucket
        fut = torch.futures.Future()
        fut.set_result(buf)
        if bucket.is_last():
            seam.pending_unsynced = False
        return fut
    W = dist.get_world_size(state.pg)
    fut = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=state.pg, async_op=True).get_future()
    last = bucket.is_last()

    def done(f):
        t = f.value()[0]
        if seam.average:
            t.div_(W)
        if last:
            seam.pending_unsynced = False
        return t

    return fut.then(done)


def install_ddp_comm_hook(ddp_model, bucket_bytes: Optional[int] = None, mode: str = "auto", comm=None, average: bool = True,
                          fp32_reduce: Optional[bool] = None) -> _SeamState:
    """`ddp_model` = the torch DistributedDataParallel accelerate.prepare built around an st355 component (module docstring)"""
    comp = getattr(ddp_model, "module", None)
    if comp is None or not hasattr(ddp_model, "register_comm_hook"):
        raise TypeError("install_ddp_comm_hook expects the torch DistributedDataParallel wrapper returned by accelerator.prepare")
    pg = getattr(ddp_model, "process_group", None)
    attach_grad_sync(comp, pg, bucket_bytes, mode, comm, fp32_reduce)
    seam = _SeamState(comp, average)
    ddp_model.register_comm_hook(_HookState(seam, pg), _comm_hook)

    def pre_forward(mod, args, kwargs=None):
        if torch.is_grad_enabled():
            seam.before_forward(bool(mod.require_backward_grad_sync))

    ddp_model.register_forward_pre_hook(pre_forward)
    ddp_model._st355_seam = seam
    return seam
