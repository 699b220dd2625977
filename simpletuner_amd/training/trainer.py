"""Minimal host harness that replays the reference's step order — `Trainer.train()` inner loop,
simpletuner/helpers/training/trainer.py:6951-7568 — around the plugin surface, so the drop-in boundary
can be exercised where SimpleTuner itself cannot be imported.  Where SimpleTuner is installed, its own
Trainer drives the same plugin / optimizer / EMA objects.

    prepare_batch -> model_predict -> loss_with_logs -> auxiliary_loss -> (gather avg loss) -> backward
      -> grad norm / clip -> optimizer.step -> zero_grad -> lr_scheduler.step -> ema.step

Differences from the reference, all on purpose (SURVEY.md §3.3 "per-step host syncs"): no per-step `.item()` — the loss
stays a device scalar and is read back only when logged; no barriers in the step; the 1/world averaging and the clip
coefficient are folded into the optimizer kernel (grad_scale) instead of extra passes over the gradients.
"""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace
from typing import Callable, Iterable, Optional

import torch
import torch.distributed as dist

from .. import ops
from .ema import EMAModel
from .grad_sync import GradSync, sync_module_states
from .multi_process import gather_sample_weighted_scalar
from .optimizer import OPTIMIZER_CHOICE, St355AdamW, St355AdamWBF16


class St355Accelerator:
    """the slice of accelerate.Accelerator the step path touches (device, ranks, backward, sync flags)."""

    def __init__(self, device=None, gradient_accumulation_steps: int = 1):
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.gradient_accumulation_steps = gradient_accumulation_steps
        self.sync_gradients = True
        self.is_dist = dist.is_available() and dist.is_initialized()
        self.process_index = dist.get_rank() if self.is_dist else 0
        self.num_processes = dist.get_world_size() if self.is_dist else 1
        self.is_main_process = self.process_index == 0

    def backward(self, loss: torch.Tensor):
        loss.backward()

    def wait_for_everyone(self):
        if self.is_dist:
            dist.barrier()


def default_config(**over) -> SimpleNamespace:
    """the config keys the drop-in honours (SURVEY.md §5 'Config / flags'), with the examples' defaults"""
    cfg = dict(model_family="flux", model_type="lora", lora_rank=32, lora_alpha=None, mixed_precision="bf16",
               weight_dtype=torch.bfloat16, base_weight_dtype=torch.bfloat16, optimizer="st355-adamw", learning_rate=1e-4,
               adam_beta1=0.9, adam_beta2=0.999, adam_epsilon=1e-8, adam_weight_decay=1e-2, use_ema=False, ema_decay=0.9999,
               ema_update_interval=None, ema_device="accelerator", ema_cpu_only=False, flow_schedule_shift=3.0,
               flow_schedule_auto_shift=False, flow_sigmoid_scale=1.0, flow_use_uniform_schedule=False, flow_use_beta_schedule=False,
               flux_fast_schedule=False, flux_guidance_mode="constant", flux_guidance_value=1.0, flux_attention_masked_training=False,
               flux_lora_target="default", snr_gamma=None, loss_type="l2", max_grad_norm=0.0, grad_clip_method="norm",
               gradient_accumulation_steps=1, train_batch_size=1, gradient_checkpointing=False, input_perturbation=0,
               offset_noise=False, seed=42, lora_init_b_std=0.0)
    cfg.update(over)
    return SimpleNamespace(**cfg)


def _safe_load(path: str, allow_pickle: bool = False):
    """torch.load with `weights_only=True` plus the numpy allow-list accelerate itself uses for its RNG-state files (accelerate.utils.other.load:
    arrays and dtypes are numbers, not code).  `allow_pickle` (config.allow_unsafe_checkpoint_pickles) is the explicit opt-in for legacy files."""
    if allow_pickle:
        return torch.load(path, map_location="cpu", weights_only=False)
    import numpy as np
    from _codecs import encode
    np_core = getattr(np, "_core", None) or np.core
    allow = [np_core.multiarray._reconstruct, np.ndarray, np.dtype, encode]
    if hasattr(np, "dtypes") and hasattr(np.dtypes, "UInt32DType"):
        allow.append(np.dtypes.UInt32DType)
    with torch.serialization.safe_globals(allow):
        return torch.load(path, map_location="cpu", weights_only=True)


def _diffusers_config(comp) -> dict:
    """a diffusers-style config.json for the trained component: `_class_name` + every JSON-representable field of `comp.config`"""
    import json
    cfg = getattr(comp, "config", None)
    items = dict(vars(cfg)) if cfg is not None and hasattr(cfg, "__dict__") else dict(cfg or {})
    out = {"_class_name": type(comp).__name__, "_diffusers_version": "0.36.0"}
    for k, v in items.items():
        if isinstance(v, tuple):
            v = list(v)
        try:
            json.dumps(v)
        except TypeError:
            continue
        out[k] = v
    return out


class Trainer:
    def __init__(self, config, model_plugin, accelerator: Optional[St355Accelerator] = None, lr_lambda: Optional[Callable] = None):
        self.config = config
        self.accelerator = accelerator or model_plugin.accelerator
        self.model = model_plugin
        comp = self.model.get_trained_component()
        # arena order when the component exposes it (fused one-launch optimizer step), else registration order
        self.params = (comp.trainable_parameters() if hasattr(comp, "trainable_parameters")
                       else [p for p in comp.parameters() if p.requires_grad])
        # optimizer_param.py:76-96 registry semantics: name -> class (+ default settings)
        opt_name = getattr(config, "optimizer", "st355-adamw")
        if opt_name not in ("adamw_bf16", "st355-adamw", "torch-adamw"):                   # never a silently different optimizer
            raise NotImplementedError(f"optimizer '{opt_name}' is not built on the st355 path (adamw_bf16, st355-adamw = torch-adamw semantics)")
        self._bf16_shadow = None
        if opt_name == "adamw_bf16":
            opt_params = self.params
            if any(p.dtype != torch.bfloat16 for p in self.params):
                opt_params = self._make_bf16_shadow()
            self.optimizer = St355AdamWBF16(opt_params, lr=config.learning_rate, betas=(config.adam_beta1, config.adam_beta2),
                                            eps=OPTIMIZER_CHOICE["adamw_bf16"]["default_settings"]["eps"], weight_decay=config.adam_weight_decay,
                                            seed=int(getattr(config, "seed", 0) or 0))
        else:
            self.optimizer = St355AdamW(self.params, lr=config.learning_rate, betas=(config.adam_beta1, config.adam_beta2),
                                        eps=config.adam_epsilon, weight_decay=config.adam_weight_decay)
        self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda) if lr_lambda else None
        if self.lr_scheduler is None and getattr(config, "lr_scheduler", None):             # the reference's named schedules (custom_schedule.py:481-557)
            from .lr_schedule import get_lr_scheduler
            self.lr_scheduler = get_lr_scheduler(config, self.optimizer, self.accelerator, None, 0)
        self.ema_model = None
        if getattr(config, "use_ema", False):
            self.ema_model = EMAModel(config, self.accelerator, self.params, decay=config.ema_decay)
        if hasattr(self.model, "configure_gradient_checkpointing"):
            self.model.configure_gradient_checkpointing()                # trainer.py:3573 / 6792
        # hip_graph: predict + loss + backward of one (shape-keyed) step are captured once into a hipGraph and replayed — the UNet step is
        # ~7000 short launches and otherwise bound by the host's launch rate.  No gradient accumulation, fixed shapes per key.  With N > 1 ranks the
        # gradient exchange runs right AFTER the replay, stream-ordered on the flat gradient buffer (the same code as the boundary step of a gradient
        # accumulation below), unless ST355_GRAPH_CAPTURE_COMM=1 and the backend can be captured (RCCL): then GradSync's collectives — issued on the
        # comm stream behind events of the capture stream — become nodes of the graph and overlap the captured backward like the eager path's do.
        self._use_graph = bool(getattr(config, "hip_graph", False))
        import os as _os
        from .ddp_seam import capture_safe
        self._graph_comm_in_graph = (self._use_graph and self.accelerator.num_processes > 1 and _os.environ.get("ST355_GRAPH_CAPTURE_COMM") == "1"
                                     and capture_safe())
        self._overlapped_sync = config.gradient_accumulation_steps == 1 and (not self._use_graph or self._graph_comm_in_graph
                                                                            or self.accelerator.num_processes == 1)
        if self.accelerator.num_processes > 1:
            sync_module_states(comp)                                     # replicas start from rank 0's weights (DDP construction semantics)
        # ST355_COMM_SINGLE_RANK=1 under an initialised process group of ONE rank: the exchange is set up and issued anyway (GradSync.single_rank_exchange) — how a
        # 1-GPU box runs this step's collectives over RCCL itself; a 1-rank SUM changes nothing, so the step's numbers are those of the plain single-process step
        single_rank = self.accelerator.num_processes == 1 and dist.is_available() and dist.is_initialized() and _os.environ.get("ST355_COMM_SINGLE_RANK", "0") not in ("", "0")
        if (self.accelerator.num_processes > 1 or single_rank) and self._overlapped_sync:
            # replicas: bucketed all-reduce of the flat gradient arena, overlapped with the hand-written backward
            import os
            comm = None
            if os.environ.get("ST355_COMM") == "native":                  # the st355_comm_* C ABI over RCCL instead of torch.distributed's collectives
                from .rccl_comm import St355Comm
                comm = St355Comm.from_process_group()
            if getattr(comp, "full", False) and getattr(comp, "grad_arena", None) is not None:
                comp.grad_sync = GradSync(comp.grad_arena, bucket_bytes=128 << 20, comm=comm)
            elif getattr(comp, "lora_grad_flat", None) is not None:
                comp.grad_sync = GradSync(comp.lora_grad_flat, comm=comm)
        if self._use_graph and config.gradient_accumulation_steps != 1:
            raise NotImplementedError("hip_graph: gradient_accumulation_steps == 1 only")
        if self._use_graph and getattr(getattr(model_plugin, "xm_config", None), "enabled", False):
            raise NotImplementedError("hip_graph: XM noise candidates read their logs on the host every step and cannot be captured")
        self._graphs = {}
        self._graph_warm = {}
        self.state = {"global_step": 0, "micro_step": 0}
        self.last_loss = None          # device scalar, no host sync
        self.last_grad_norm = None
        self.last_grad_absmax = None

    def _make_bf16_shadow(self):
        """adamw_bf16 over an fp32 adapter arena (LoRA): the reference's example trains bf16 adapter weights with bf16 gradients under AdamWBF16
        (simpletuner/examples/sd3.peft-lora/config.json: optimizer adamw_bf16, mixed_precision bf16; optimizers/adamw_bfloat16/__init__.py:66 asserts bf16).
        The trained values live in a bf16 arena that the optimizer steps (compensated stochastic-rounding update, its `shift` buffer); the engine's fp32
        arena mirrors it exactly (every value is a bf16 number), and the fp32 rank-space gradients are rounded to bf16 once per step — what autograd hands
        a bf16 parameter.  Returns the bf16 parameters the optimizer owns (same order, same shapes)."""
        from .optimizer import _contiguous_run
        if not _contiguous_run([p.data for p in self.params]):
            raise NotImplementedError("adamw_bf16 over fp32 trainables needs them as one flat arena (the LoRA adapter arena)")
        n = sum(p.numel() for p in self.params)
        flat32 = torch.as_strided(self.params[0].data, (n,), (1,))
        master = flat32.to(torch.bfloat16)
        flat32.copy_(master)                                  # the engine computes with exactly the values the optimizer holds
        grad16 = torch.zeros(n, dtype=torch.bfloat16, device=flat32.device)
        shadow, off = [], 0
        for p in self.params:
            k = p.numel()
            q = torch.nn.Parameter(master[off:off + k].view_as(p))
            q.grad = grad16[off:off + k].view_as(p)
            shadow.append(q)
            off += k
        self._bf16_shadow = SimpleNamespace(flat32=flat32, master=master, grad16=grad16, params=shadow, n=n)
        return shadow

    def check_pending_loss(self) -> None:
        """The reference raises `RuntimeError("Non-finite training loss detected …")` right after the loss is computed (trainer.py:7102-7110), which
        costs a host sync in the middle of every step.  Here the check is opt-in (`config.check_finite_loss`) and DEFERRED: the loss of step i is
        inspected at the start of step i+1 (or by calling this), when it has long been computed — same exception, same message fields, one step
        late, no stall of the launch queue."""
        pending, self._pending_loss = getattr(self, "_pending_loss", None), None
        if pending is None:
            return
        loss, step, filepaths, backend = pending
        if not bool(torch.isfinite(loss).all()):
            raise RuntimeError(f"Non-finite training loss detected (loss={loss.item()}, data_backend_id={backend}, filepaths={filepaths}, "
                               f"loss_logs={{}}) at optimizer step {step}.")

    def train_step(self, raw_batch: dict) -> torch.Tensor:
        cfg, acc = self.config, self.accelerator
        if getattr(cfg, "check_finite_loss", False):
            self.check_pending_loss()
        comp = self.model.get_trained_component()
        prepared = self.model.prepare_batch(raw_batch, self.state)                       # trainer.py:6964
        self.state["micro_step"] += 1
        boundary = self.state["micro_step"] % cfg.gradient_accumulation_steps == 0       # accelerator.accumulate (:7009)
        acc.sync_gradients = boundary
        sync = getattr(comp, "grad_sync", None)
        if sync is not None:
            sync.enabled = boundary
        if self._use_graph:
            loss = self._graph_forward_backward(prepared)
            self.last_loss = gather_sample_weighted_scalar(loss, prepared["latents"].shape[0], acc)
        else:
            pred = self.model.model_predict(prepared)                                    # :7097 -> :6085
            loss, _ = self.model.loss_with_logs(prepared, pred)
            loss, _ = self.model.auxiliary_loss(pred, prepared, loss)
            if cfg.gradient_accumulation_steps > 1:
                loss = loss / cfg.gradient_accumulation_steps
            self.last_loss = gather_sample_weighted_scalar(loss, prepared["latents"].shape[0], acc)   # :7114 (C2)
            acc.backward(loss)                                                           # :7126
        if getattr(cfg, "check_finite_loss", False):
            self._pending_loss = (loss.detach(), self.state["global_step"], prepared.get("filepaths"), prepared.get("data_backend_id"))
        if not boundary:
            return self.last_loss
        grad_scale = getattr(comp, "grad_scale_from_sync", 1.0) if sync is not None else 1.0
        if acc.num_processes > 1 and not self._overlapped_sync:
            # gradient accumulation: reduce the ACCUMULATED .grad once at the boundary (DDP no_sync semantics, trainer.py:7009)
            from .optimizer import _contiguous_run
            grads = [p.grad for p in self.params]
            if not _contiguous_run(grads):
                raise RuntimeError("accumulated gradients are not one flat arena")
            n = sum(g.numel() for g in grads)
            dist.all_reduce(torch.as_strided(grads[0], (n,), (1,)), op=dist.ReduceOp.SUM)
            grad_scale = 1.0 / acc.num_processes
        if getattr(cfg, "max_grad_norm", 0) and cfg.max_grad_norm > 0:                   # :7138-7217
            # the flat view is built from the gradients autograd actually holds: with gradient accumulation AccumulateGrad adds later micro-steps IN
            # PLACE into the first micro-step's buffer, so `p.grad` (not the component's last backward buffer) is what gets clipped, reduced and stepped
            from .optimizer import _contiguous_run
            grads = [p.grad for p in self.params]
            if any(g is None for g in grads) or not _contiguous_run(grads):
                raise NotImplementedError("gradient clipping needs the trainable gradients as one flat arena (every parameter with a gradient, back to back)")
            gflat = torch.as_strided(grads[0], (sum(g.numel() for g in grads),), (1,))
            method = getattr(cfg, "grad_clip_method", "norm")
            if method == "value":                                                        # accelerator.clip_grad_value_ (:7209-7213)
                ops.grad_clamp_(gflat, cfg.max_grad_norm / grad_scale)                   # gradients are rank SUMS here; 1/world lives in grad_scale
            elif method == "norm":
                stats = ops.grad_norm(gflat)
                self.last_grad_norm = stats[0].sqrt() * grad_scale                       # device scalars, read only when logged
                self.last_grad_absmax = stats[1] * grad_scale          # `_max_grad_value` (trainer.py:6376-6407): same pass
                ops.grad_clip_norm_(gflat, stats, cfg.max_grad_norm, pre_scale=grad_scale)   # coefficient computed and applied on the device: no host sync
            else:
                raise ValueError(f"Unknown grad clip method: {method}. Supported methods: value, norm")
        self.optimizer.grad_scale = grad_scale
        ema_fused = False
        sh = self._bf16_shadow
        if sh is not None:                                                               # adamw_bf16 over the fp32 adapter arena: bf16 gradients in, bf16 weights out
            from .optimizer import _contiguous_run
            grads = [p.grad for p in self.params]
            if any(g is None for g in grads) or not _contiguous_run(grads):
                raise RuntimeError("adamw_bf16 (LoRA): the adapter gradients are not one flat arena")
            sh.grad16.copy_(torch.as_strided(grads[0], (sh.n,), (1,)))
            if any(q.grad is None for q in sh.params):      # (a zero_grad(set_to_none=True) by the caller must not detach the optimizer's parameters from their gradient arena)
                off = 0
                for q in sh.params:
                    q.grad = sh.grad16[off:off + q.numel()].view_as(q)
                    off += q.numel()
            self.optimizer.step()
            sh.flat32.copy_(sh.master)
            if not self._use_graph:
                for p in self.params:
                    p.grad = None
        else:
            # EMA fused into the optimizer's launch (north star: "fused AdamW, EMA update"): the decay of ema_model.step(params, global_step + 1) depends on the
            # step count alone (ema.py:322-349), so it is known here; the kernel applies s -= (1 - d)(s - p_new) to the element it just updated — the same
            # arithmetic as the separate pass (ema.py:393-433), minus its 3 x 2 B/param of HBM traffic and one launch
            ema_decay = None
            if self.ema_model is not None and isinstance(self.optimizer, St355AdamW):
                ema_decay = self.ema_model.fused_decay(self.params, self.state["global_step"] + 1)
            if ema_decay is not None:
                self.optimizer.ema_shadow_flat, self.optimizer.ema_decay, self.optimizer.ema_applied = self.ema_model.shadow_flat, float(ema_decay), False
            self.optimizer.step()                                                        # :7239
            if ema_decay is not None:
                self.optimizer.ema_shadow_flat = None
                ema_fused = bool(self.optimizer.ema_applied)
            if not self._use_graph:                                                      # graph mode: the captured backward re-writes the same .grad tensors
                self.optimizer.zero_grad(set_to_none=True)                               # :7253
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()                                                     # :7293
        self.state["global_step"] += 1
        if self.ema_model is not None:
            if ema_fused:
                self.ema_model.commit_fused(self.state["global_step"], ema_decay)
            else:
                self.ema_model.step(self.params, self.state["global_step"])              # :7352
        return self.last_loss

    # ------------------------------------------------------------------------------------------------
    def _eager_forward_backward(self, prepared):
        pred = self.model.model_predict(prepared)
        loss, _ = self.model.loss_with_logs(prepared, pred)
        loss, _ = self.model.auxiliary_loss(pred, prepared, loss)
        self.accelerator.backward(loss)
        return loss

    def release_graphs(self) -> None:
        """drop every captured step and the shared capture pool (their activations go back to the allocator): before eager steps that need the memory"""
        import gc
        self._graphs.clear()
        self._graph_pool = None
        gc.collect()
        torch.cuda.empty_cache()

    def _graph_forward_backward(self, prepared):
        def tensors(d, prefix=""):
            for k, v in d.items():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    yield prefix + k, v
                elif isinstance(v, dict):
                    yield from tensors(v, prefix + k + "/")

        def clone_tree(d):
            return {k: (v.clone() if isinstance(v, torch.Tensor) and v.is_cuda else clone_tree(v) if isinstance(v, dict) else v) for k, v in d.items()}

        def shallow(d):          # plugins may replace entries of the batch dict (Flux divides `timesteps` by 1000): never let that touch `static`
            return {k: (shallow(v) if isinstance(v, dict) else v) for k, v in d.items()}

        key = tuple((k, tuple(v.shape), v.dtype) for k, v in tensors(prepared))
        entry = self._graphs.get(key)
        if entry is None:
            # warm-up ON A SIDE STREAM (workspaces, kernel attributes, allocator pools, and AccumulateGrad nodes bound to the capture-side stream:
            # a node created on the default stream and kept alive breaks capture), then capture one step
            static = clone_tree(prepared)
            static_inputs = dict(tensors(static))          # name -> the tensors the captured kernels read; filled from each step's batch
            self.last_loss = None
            # ONE side stream for every capture: the caching allocator keys its free blocks by stream, so a fresh stream per bucket shape made every capture take a
            # new ~40 GB of the shared pool instead of the blocks the previous capture had freed (r6 probe: reserved +25 GiB per bucket at batch 4, allocated flat)
            if getattr(self, "_graph_stream", None) is None:
                self._graph_stream = torch.cuda.Stream()
            side = self._graph_stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    for p in self.params:
                        p.grad = None
                    loss = self._eager_forward_backward(shallow(static))
                    del loss
            torch.cuda.current_stream().wait_stream(side)
            for p in self.params:
                p.grad = None
            torch.cuda.synchronize()
            torch.cuda.empty_cache()               # the eager warm-up's cached blocks go back before the capture builds its own pool
            if os.environ.get("ST355_GRAPH_MEM_DEBUG"):
                print(f"[graph] before capture: allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, "
                      f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", file=sys.stderr)
            g = torch.cuda.CUDAGraph()
            # one memory pool for every captured shape (mixed aspect buckets: one graph per bucket): the graphs never run concurrently, so the activations of
            # one capture are the next capture's free blocks — five private ~40 GB pools did not fit next to the model (r5), one shared pool does
            if getattr(self, "_graph_pool", None) is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(g, pool=self._graph_pool, stream=side):      # the SAME stream as the warm-up: the parameters' AccumulateGrad nodes are bound to it
                loss = self._eager_forward_backward(shallow(static))
            entry = self._graphs[key] = (g, static_inputs, loss.detach(), [p.grad for p in self.params])
        g, st, loss, grads = entry
        for k, v in tensors(prepared):
            st[k].copy_(v)
        g.replay()
        for p, gr in zip(self.params, grads):
            p.grad = gr
        return loss.clone()        # the static tensor is overwritten by the next replay: callers keep their own value

    # ---- checkpoints: the files of the reference's save hooks + accelerate.save_state, in one directory (save_hooks.py:368-443, 850-1305) ----
    def save_state(self, checkpoint_dir: str) -> None:
        """`<dir>/optimizer.bin`, `scheduler.bin`, `random_states_<rank>.pkl` (accelerate's names and pickled keys), the trained weights
        (`pytorch_lora_weights.safetensors` for adapters, `<subfolder>/diffusion_pytorch_model.safetensors` for a full fine-tune),
        `<subfolder>_ema/ema_model.pt`, `training_state.json` (StateTracker.save_training_state keys) and the round-robin timestep cursor.
        Rank 0 writes the shared files; every rank writes its own RNG / cursor files."""
        import json
        import os
        import random

        import numpy as np
        acc, plug = self.accelerator, self.model
        rank = int(getattr(acc, "process_index", 0) or 0)
        os.makedirs(checkpoint_dir, exist_ok=True)
        rng = {"random_state": random.getstate(), "numpy_random_seed": np.random.get_state(), "torch_manual_seed": torch.get_rng_state(),
               "torch_cuda_manual_seed": torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None,
               "st355_noise_step": int(getattr(plug, "_noise_step", 0)),
               "st355_noise_offset": int(getattr(plug, "_noise_offset", 0))}    # the in-kernel Philox counter position of the noising pass (cumulative, shape-independent)
        torch.save(rng, os.path.join(checkpoint_dir, f"random_states_{rank}.pkl"))     # accelerate's save_accelerator_state: torch.save under the .pkl name
        plug.save_flow_custom_timestep_state(checkpoint_dir)
        ts = {"global_step": self.state["global_step"], "epoch_step": self.state.get("epoch_step", 0), "epoch": self.state.get("epoch", 1),
              "exhausted_backends": list(self.state.get("exhausted_backends", [])), "repeats": dict(self.state.get("repeats", {})),
              "micro_step": self.state["micro_step"]}
        if rank != 0:                                                                   # save_hooks.py:369-373: non-zero ranks write their own state file
            with open(os.path.join(checkpoint_dir, f"training_state-rank{rank}.json"), "w") as fh:
                json.dump(ts, fh)
        if rank != 0:
            return
        host = lambda sd: {"state": {k: {n: (v.detach().to("cpu") if torch.is_tensor(v) else v) for n, v in st.items()} for k, st in sd["state"].items()},
                           "param_groups": sd["param_groups"]}
        torch.save(host(self.optimizer.state_dict()), os.path.join(checkpoint_dir, "optimizer.bin"))
        if self.lr_scheduler is not None:
            torch.save(self.lr_scheduler.state_dict(), os.path.join(checkpoint_dir, "scheduler.bin"))
        comp = plug.get_trained_component()
        if getattr(comp, "full", False) and hasattr(comp, "diffusers_state_dict"):
            from safetensors.torch import save_file
            sub = os.path.join(checkpoint_dir, plug.MODEL_SUBFOLDER)
            os.makedirs(sub, exist_ok=True)
            save_file({k: v.detach().to("cpu").contiguous() for k, v in comp.diffusers_state_dict().items()},
                      os.path.join(sub, "diffusion_pytorch_model.safetensors"), metadata={"format": "pt"})
            # `unwrapped_model.save_pretrained` (save_hooks.py:1128) also writes config.json: without it neither diffusers' from_pretrained nor the
            # reference's resume path can open the folder
            with open(os.path.join(sub, "config.json"), "w") as fh:
                json.dump(_diffusers_config(comp), fh, indent=2, sort_keys=True)
        else:
            plug.save_lora_weights(checkpoint_dir)
        if self.ema_model is not None:
            self.ema_model.save_state_dict(os.path.join(checkpoint_dir, f"{plug.MODEL_SUBFOLDER}_ema", "ema_model.pt"))
        with open(os.path.join(checkpoint_dir, "training_state.json"), "w") as fh:
            json.dump(ts, fh)

    def load_state(self, checkpoint_dir: str) -> None:
        """the inverse of save_state; a missing per-rank RNG file falls back to rank 0's (resuming on more GPUs than the run was saved with)"""
        import json
        import os
        import random

        import numpy as np
        acc, plug = self.accelerator, self.model
        rank = int(getattr(acc, "process_index", 0) or 0)
        comp = plug.get_trained_component()
        full_path = os.path.join(checkpoint_dir, plug.MODEL_SUBFOLDER, "diffusion_pytorch_model.safetensors")
        if getattr(comp, "full", False) and os.path.exists(full_path):
            from safetensors.torch import load_file
            comp.load_diffusers_state(load_file(full_path))
        else:
            plug.load_lora_weights(input_dir=checkpoint_dir)
            sh = self._bf16_shadow
            if sh is not None:                    # adamw_bf16 over the adapter arena: the optimizer's bf16 values follow the loaded weights (the saved adapters ARE bf16 numbers
                sh.master.copy_(sh.flat32)        # when this trainer wrote them; a foreign file is rounded once, as at construction), and the engine's arena mirrors them
                sh.flat32.copy_(sh.master)
        unsafe = bool(getattr(self.config, "allow_unsafe_checkpoint_pickles", False))     # explicit opt-in only: the default never executes a pickle
        self.optimizer.load_state_dict(_safe_load(os.path.join(checkpoint_dir, "optimizer.bin"), unsafe))
        sched = os.path.join(checkpoint_dir, "scheduler.bin")
        if self.lr_scheduler is not None and os.path.exists(sched):
            self.lr_scheduler.load_state_dict(_safe_load(sched, unsafe))
        if self.ema_model is not None:
            self.ema_model.load_state_dict(os.path.join(checkpoint_dir, f"{plug.MODEL_SUBFOLDER}_ema", "ema_model.pt"))
        ts_path = os.path.join(checkpoint_dir, "training_state.json" if rank == 0 else f"training_state-rank{rank}.json")
        if not os.path.exists(ts_path):                                                  # save_hooks.py:1276-1279: fall back to the default name
            ts_path = os.path.join(checkpoint_dir, "training_state.json")
        with open(ts_path) as fh:
            ts = json.load(fh)
        self.state.update({k: ts[k] for k in ("global_step", "epoch_step", "epoch", "exhausted_backends", "repeats") if k in ts})
        self.state["micro_step"] = int(ts.get("micro_step", self.state["global_step"] * self.config.gradient_accumulation_steps))
        plug.load_flow_custom_timestep_state(checkpoint_dir, fallback_global_step=self.state["global_step"])
        for name in (f"random_states_{rank}.pkl", "random_states_0.pkl"):
            path = os.path.join(checkpoint_dir, name)
            if not os.path.exists(path):
                continue
            rng = _safe_load(path, unsafe)
            random.setstate(rng["random_state"])
            np.random.set_state(rng["numpy_random_seed"])
            torch.set_rng_state(rng["torch_manual_seed"])
            if rng.get("torch_cuda_manual_seed") is not None and torch.cuda.is_available():
                torch.cuda.set_rng_state_all(rng["torch_cuda_manual_seed"])
            plug._noise_step = int(rng.get("st355_noise_step", 0))
            plug._noise_offset = int(rng.get("st355_noise_offset", 0))
            break

    def train(self, batches: Iterable[dict], max_steps: int):
        losses = []
        for i, b in enumerate(batches):
            if i >= max_steps:
                break
            losses.append(self.train_step(b))
        return losses
