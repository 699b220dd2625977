"""St355Comm — the `st355_comm_*` C-ABI entry points (include/st355.h; csrc/comm.hip) as a Python object: RCCL collectives on raw device pointers
and the caller's HIP stream, no torch.distributed on the data path.

`GradSync(..., comm=St355Comm.from_process_group())` (or env `ST355_COMM=native` in `Trainer`) routes the gradient exchange of the replicas
through it; the default stays torch.distributed's RCCL backend (same library underneath, one more well-trodden layer).  torch.distributed is used
here only as the side channel that carries the 128-byte unique id from rank 0 to the other ranks — exactly the job the header leaves to the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import lib as _l

_KIND = {torch.float32: 0, torch.bfloat16: 1}


class St355Comm:
    def __init__(self, unique_id: bytes, world: int, rank: int):
        if len(unique_id) != 128:
            raise ValueError("St355Comm: the unique id is 128 bytes (st355_comm_unique_id)")
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        _l.check(_l.load().st355_comm_init(C.byref(self._h), C.c_char_p(unique_id), self.world, self.rank), "comm_init")

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _l.check(_l.load().st355_comm_unique_id(buf), "comm_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None) -> "St355Comm":
        """rank 0 draws the id; torch.distributed (any backend) broadcasts the 128 bytes; every rank builds its communicator on its current device"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return cls(cls.unique_id(), 1, 0)
        box = [cls.unique_id() if dist.get_rank(group) == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], dist.get_world_size(group), dist.get_rank(group))

    def _args(self, t: torch.Tensor):
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in _KIND:
            raise _l.St355Error("St355Comm: expected a contiguous fp32 / bf16 device tensor")
        return t.data_ptr(), _KIND[t.dtype], torch.cuda.current_stream(t.device).cuda_stream

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        p, k, s = self._args(t)
        _l.check(_l.load().st355_comm_all_reduce(self._h, s, p, t.numel(), k), "comm_all_reduce")
        return t

    def reduce_scatter_(self, seg: torch.Tensor) -> torch.Tensor:
        """in place on `seg` (numel a multiple of world): this rank's shard seg[rank*n/W : (rank+1)*n/W] receives the SUM over ranks; returns the shard view"""
        p, k, s = self._args(seg)
        n = seg.numel() // self.world
        if n * self.world != seg.numel():
            raise _l.St355Error("St355Comm.reduce_scatter_: numel must be a multiple of the world size")
        shard = seg[self.rank * n:(self.rank + 1) * n]
        _l.check(_l.load().st355_comm_reduce_scatter(self._h, s, p, shard.data_ptr(), n, k), "comm_reduce_scatter")
        return shard

    def all_gather_(self, seg: torch.Tensor) -> torch.Tensor:
        """in place on `seg`: every rank's shard (as left by reduce_scatter_) is gathered into all of seg"""
        p, k, s = self._args(seg)
        n = seg.numel() // self.world
        shard = seg[self.rank * n:(self.rank + 1) * n]
        _l.check(_l.load().st355_comm_all_gather(self._h, s, shard.data_ptr(), p, n, k), "comm_all_gather")
        return seg

    def destroy(self):
        if self._h:
            _l.check(_l.load().st355_comm_destroy(self._h), "comm_destroy")
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
