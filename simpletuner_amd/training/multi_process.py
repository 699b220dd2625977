"""Rank helpers + the step loop's small collectives (C2, C3), mirroring
simpletuner/helpers/data_backend/runtime/context_parallel_sync.py:327-348 and simpletuner/helpers/training/trainer.py:423-432.
Backend-agnostic: `nccl` (= RCCL over xGMI on ROCm) on GPUs, `gloo` in the CPU tests."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def gather_sample_weighted_scalar(value: torch.Tensor, local_batch_size: int, accelerator=None) -> torch.Tensor:
    """sum_r(loss_r * b_r) / sum_r(b_r): one fixed-shape [2] contribution per rank (context_parallel_sync.py:327-348)."""
    local_batch_size = int(local_batch_size)
    if local_batch_size < 1:
        raise ValueError("local_batch_size must be greater than 0.")
    if value.numel() != 1:
        raise ValueError("Sample-weighted scalar gather requires a scalar tensor.")
    value = value.detach().float().reshape(())
    ws = world_size()
    if ws == 1:
        return value
    contribution = torch.stack((value * float(local_batch_size), value.new_tensor(float(local_batch_size))))
    gathered = [torch.empty_like(contribution) for _ in range(ws)]
    dist.all_gather(gathered, contribution)
    totals = torch.stack(gathered).reshape(-1, 2).sum(dim=0)
    return totals[0] / totals[1]


def any_rank_reached_epoch_end(flag: bool, device) -> bool:
    """all-reduce(MAX) of a 1-int flag (trainer.py:423-432)"""
    if world_size() == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item())
