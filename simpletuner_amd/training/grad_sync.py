"""Data-parallel gradient synchronisation for replicas (C1): what DDP's reducer does in the reference
(trainer.py:1034-1041, 4564-4571), re-designed for MI355X:

  * gradients live in ONE flat arena, filled back-to-front as the hand-written backward walks the blocks in reverse;
  * every time a bucket's worth of finished gradient has accumulated, the slice is handed to the COMM STREAM: an event recorded on
    the compute stream marks "these gradient bytes are final", the comm stream waits for that event (device-side, the host never
    blocks) and the collective is issued there — RCCL over xGMI with the `nccl` backend, `gloo` in the CPU / shared-GPU tests — so it
    overlaps the remaining backward kernels; `finish()` joins the comm stream back into the compute stream with one event;
  * two forms of the exchange (same result, SUM over ranks):
      - `allreduce`: one all-reduce per bucket — right for the small LoRA arenas (Flux r32: 150 MB fp32, ~0.25 ms of link time);
      - `rs_ag`: reduce-scatter of the bucket into this rank's 1/N shard, then all-gather of the shards, issued back to back on the
        comm stream — the direct form for the xGMI full mesh (each of the 7 links carries 2*G/N bytes), used for the
        4-5 GB full-fine-tune arenas (default: arenas >= 1 GiB).  The shard boundary is where a sharded optimizer / a shard-local
        gradient-norm would slot in; both collectives run in place on the arena (recv = send + rank * count).
  * SUM is used and the 1/world_size averaging is folded into the optimizer kernel's grad_scale (no extra pass over HBM);
  * `no_sync()` mirrors DDP/accelerate semantics for gradient accumulation (trainer.py:7009);
  * `fp32_reduce` (opt-in for bf16 arenas in the `rs_ag` form, ST355_FP32_REDUCE=1): the reduce-scatter half becomes an ALL-TO-ALL of the bf16 chunks — chunk j of every
    rank travels straight to rank j over its own xGMI link, the same (N-1)/N * G bytes per rank as a reduce-scatter — followed by a local sum of the
    N received chunks in fp32, in rank order (st355_sum_chunks_bf16: deterministic, ONE rounding to bf16), then the all-gather of the reduced shards.
    A bf16 RCCL SUM rounds after every hop of its ring / tree; this form accumulates the 2-billion-element full-fine-tune gradient in fp32
    without putting fp32 on the wire;
  * `ST355_COMM_TIMING=1`: every slice's collective is bracketed by timing events on the comm stream and the backward by events on the compute stream;
    `overlap_report()` turns them into per-bucket timestamps, the exposed tail and the overlap fraction (what a SCALE run reports).
Bucket size is chosen for the per-link xGMI bound: 32 MiB slices keep each of the 7 links busy for ~0.2 ms (2*G/N per link at
~153 GB/s), long enough to amortise launch latency, short enough to overlap.
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch
import torch.distributed as dist

RS_AG_MIN_BYTES = 1 << 30


def hand_over_gradients(model, arena: torch.Tensor) -> torch.Tensor:
    """What every st355 backward does last: give autograd a PRIVATE flat gradient buffer — a copy of the arena, or the arena itself when the model double-buffers
    it (`_swap_grad_arena`) — (so `.grad` never aliases the buffer the next backward
    overwrites; the per-parameter gradients stay views of one contiguous tensor, which the fused optimizer and the exchange exploit).  Under a DDP-style
    wrapper (training.ddp_seam) the same pass applies 1/world_size — DDP hands out AVERAGED gradients — and the wrapper's end-of-backward callback is queued
    on the autograd engine."""
    scale = getattr(model, "_handover_scale", None)
    swap = getattr(model, "_swap_grad_arena", None)
    if scale is None and swap is not None and getattr(model, "grad_arena", None) is arena:
        # double-buffered arena (sd3 full fine-tune): the arena goes to autograd as it is and the model's next backward fills the other one — no 4 GB clone per step
        gflat = arena
        swap()
    else:
        gflat = arena.clone() if scale is None else arena * scale
    cb = getattr(model, "_post_backward_cb", None)
    if cb is not None:
        torch.autograd.Variable._execution_engine.queue_callback(cb)
    return gflat


class GradSync:
    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 32 << 20, process_group=None, mode: str = "auto", comm=None,
                 fp32_reduce: Optional[bool] = None, single_rank_exchange: Optional[bool] = None):
        if mode not in ("auto", "allreduce", "rs_ag"):
            raise ValueError(f"GradSync mode {mode!r}: expected auto / allreduce / rs_ag")
        self.flat = flat_grad
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())
        self.pg = process_group
        self.comm = comm                 # optional training.rccl_comm.St355Comm: the exchange goes through the st355_comm_* C ABI instead of torch.distributed
        self.enabled = True
        if mode == "auto":
            mode = "rs_ag" if flat_grad.numel() * flat_grad.element_size() >= RS_AG_MIN_BYTES else "allreduce"
        self.mode = mode
        self._works: List = []
        self._pending: List[List[int]] = []   # finished-but-unsent regions [lo, hi), disjoint, in the order they were opened (adjacent ready() ranges merge into them)
        # no collective is issued over more than max_slice_elems elements (ST355_COMM_MAX_SLICE_MB, default 256 MiB): a region that grew past it — the 1.5 GB front of the
        # SD3-Medium arena (embedders + the fused modulation matrix) — goes out as consecutive sub-slices.  Found on the MI355X in r06
        # (tools/probes/rccl_large_slice_probe.py): all_to_all_single over RCCL 2.26.6 delivers only the first HALF of a per-peer chunk larger than 1 GiB (a 1531 MiB
        # chunk arrived intact up to 766 MiB, a 2 GiB one up to 1 GiB; all-reduce / reduce-scatter / all-gather were right at every size tried), so the
        # fp32-accumulating form silently summed garbage for the tail of such a slice.  The cap also bounds the receive buffer of that form.
        env_cap = os.environ.get("ST355_COMM_MAX_SLICE_MB")
        self.max_slice_elems = max(1, (int(env_cap) if env_cap else 256) * (1 << 20) // flat_grad.element_size())
        self.launched_slices: List = []  # (lo, hi) of every exchange issued in the current backward (tests inspect it)
        self.launched_ops: List = []     # ("all_reduce" | "reduce_scatter" | "all_gather", lo, hi) in issue order
        # explicit comm stream (device arenas only): collectives are enqueued behind an event of the compute stream, never on it
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if flat_grad.is_cuda else None
        self._comm_used = False
        self._serial_backend = None      # resolved lazily: gloo runs async works concurrently -> dependent collectives must be waited for
        # fp32-accumulating reduce-scatter for bf16 arenas in the rs_ag form (module docstring): OPT-IN (fp32_reduce=True or ST355_FP32_REDUCE=1) until the
        # all-to-all + st355_sum_chunks_bf16 + all-gather sequence has run over RCCL with more than one rank (no multi-GPU box has executed this repo yet; the
        # kernel itself is checked on one GPU by tests/test_kernels_gpu.py::test_sum_chunks_bf16); the C-ABI comm path keeps RCCL's own reduce-scatter
        env = os.environ.get("ST355_FP32_REDUCE")
        default = bool(env and env != "0") and flat_grad.dtype == torch.bfloat16
        self.fp32_reduce = default if fp32_reduce is None else bool(fp32_reduce)
        self._recv: Optional[torch.Tensor] = None
        self._fp32_fallback_logged = False
        # a world of ONE normally exchanges nothing.  single_rank_exchange (or ST355_COMM_SINGLE_RANK=1) issues every collective anyway: on a 1-GPU box this is
        # the only way the stream-ordered branch below (async work handles parked until finish(), the comm-stream join, the all-to-all + fp32 sum form) runs over
        # RCCL itself rather than over gloo — a 1-rank SUM leaves the arena as it was, so the step's numbers must not move (tests/test_distributed_gpu.py)
        env1 = os.environ.get("ST355_COMM_SINGLE_RANK")
        self.single_rank_exchange = bool(env1 and env1 != "0") if single_rank_exchange is None else bool(single_rank_exchange)
        self.timing = bool(os.environ.get("ST355_COMM_TIMING")) and flat_grad.is_cuda
        self._ev_begin = self._ev_end = None
        self._ev_slices: List = []

    @property
    def world_size(self) -> int:
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.pg) if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        if self.comm is not None:
            return self.comm.rank
        return dist.get_rank(self.pg) if dist.is_available() and dist.is_initialized() else 0

    def begin(self):
        self._works, self._pending, self.launched_slices, self.launched_ops, self._comm_used = [], [], [], [], False
        if self.timing:
            self._ev_begin = torch.cuda.Event(enable_timing=True)
            self._ev_begin.record(torch.cuda.current_stream(self.flat.device))
            self._ev_slices, self._ev_end = [], None

    def _stream_ordered(self) -> bool:
        """True when the backend executes this group's collectives in issue order on one device stream (nccl == RCCL)"""
        if self._serial_backend is None:
            self._serial_backend = str(dist.get_backend(self.pg)).lower() == "nccl"
        return self._serial_backend

    def _fp32_alltoall_ok(self) -> bool:
        """the all-to-all + local fp32 sum + all-gather form needs the three steps ordered: a device arena under a stream-ordered backend (RCCL), or a host arena
        under gloo (whose blocking all-to-all orders them)"""
        return self.flat.is_cuda == self._stream_ordered()

    def _comm_ctx(self):
        if self.comm_stream is None:
            return contextlib.nullcontext()
        # the gradient bytes of this slice are final at this point of the backward: the comm stream waits for the compute stream's work enqueued so far.
        # wait_stream, NOT a temporary torch.cuda.Event + wait_event: the Python event died at the end of this function while the wait was still queued, and
        # once the host ran far enough ahead of the device (the block-level C entry points: fewer host calls per block) the comm stream's barrier referenced a
        # destroyed signal — a GPU memory-access fault in the first step of every 2-rank run (r3: 7 of 7 runs; 0 of 6 with this form; gpurun_out/dbg_n2_*)
        self.comm_stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        self._comm_used = True
        return torch.cuda.stream(self.comm_stream)

    def _fire(self, lo: int, hi: int):
        """hand [lo, hi) to the exchange, in sub-slices of at most max_slice_elems (cut on multiples of 8 * world elements from `lo`, so that an 8-aligned region keeps
        every sub-slice eligible for the fp32-accumulating form)"""
        W = max(1, self.world_size)
        step = max(8 * W, self.max_slice_elems // (8 * W) * (8 * W))
        while hi - lo > step:
            self._fire_one(lo, lo + step)
            lo += step
        self._fire_one(lo, hi)

    def _fire_one(self, lo: int, hi: int):
        if hi <= lo:
            return
        self.launched_slices.append((lo, hi))
        W = self.world_size
        if not self.enabled or W < 1 or (W == 1 and not (self.single_rank_exchange and (self.comm is not None or (dist.is_available() and dist.is_initialized())))):
            return
        with self._comm_ctx():
            ev0 = None
            if self.timing:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record(self.comm_stream)
            self._fire_on_comm_stream(lo, hi, W)
            if ev0 is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record(self.comm_stream)
                self._ev_slices.append((lo, hi, ev0, ev1))

    @staticmethod
    def fp32_span(lo: int, hi: int, W: int) -> int:
        """elements of [lo, hi) the fp32-accumulating reduce-scatter may take: st355_sum_chunks_bf16 needs every rank's chunk to be a multiple of 8 elements
        that starts on a 16-byte boundary — so the span is a multiple of 8 * W and `lo` itself must be 8-aligned (arena tensors are padded to 8 elements, so a
        region start normally is); otherwise 0 and the slice takes RCCL's own reduce-scatter.  The up-to 8 W - 1 leftover elements ride the tail all-reduce."""
        if lo % 8 != 0 or W <= 0:
            return 0
        return (hi - lo) // (8 * W) * (8 * W)

    def _sum_chunks(self, recv: torch.Tensor, W: int, out: torch.Tensor):
        """out[i] = bf16(sum over w, in rank order, of float(recv[w, i])) — the local half of the fp32-accumulating reduce-scatter"""
        if recv.is_cuda:
            from .. import ops
            ops.sum_chunks_bf16(recv, W, out)
        else:                                                          # gloo / CPU plumbing tests
            out.copy_(recv.view(W, -1).to(torch.float32).sum(dim=0))

    def _fire_on_comm_stream(self, lo: int, hi: int, W: int):
            m = (hi - lo) // W * W if self.mode == "rs_ag" else 0
            m8 = self.fp32_span(lo, hi, W) if self.mode == "rs_ag" else 0
            if self.fp32_reduce and self.mode == "rs_ag" and m > 0 and m8 == 0 and not self._fp32_fallback_logged:
                self._fp32_fallback_logged = True        # say it once: a requested fp32-accumulating reduce must not degrade silently
                import logging
                logging.getLogger("st355.grad_sync").warning(
                    "fp32_reduce requested, but the slice [%d, %d) cannot take it (start not 8-element aligned or shorter than 8 x world): "
                    "this slice uses the backend's own reduce-scatter in the arena dtype", lo, hi)
            if m8 > 0 and self.fp32_reduce and self.comm is None and self.flat.dtype == torch.bfloat16 and self._fp32_alltoall_ok():
                m = m8                                                 # st355_sum_chunks_bf16 wants 8-element (16-byte) chunks: the tail below takes the rest
                seg = self.flat[lo:lo + m]
                if self._recv is None or self._recv.numel() < m:
                    self._recv = torch.empty(m, dtype=seg.dtype, device=seg.device)
                recv = self._recv[:m]
                shard = seg[self.rank * (m // W):(self.rank + 1) * (m // W)]
                dist.all_to_all_single(recv, seg, group=self.pg)       # chunk j of every rank -> rank j (point-to-point over xGMI; stream-ordered under RCCL)
                self.launched_ops.append(("all_to_all", lo, lo + m))
                self._sum_chunks(recv, W, shard)
                self._works.append(dist.all_gather_into_tensor(seg, shard, group=self.pg, async_op=True))
                self.launched_ops.append(("all_gather", lo, lo + m))
                if lo + m < hi:
                    self._works.append(dist.all_reduce(self.flat[lo + m:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                    self.launched_ops.append(("all_reduce", lo + m, hi))
                return
            if self.comm is not None:                                  # C-ABI RCCL path: stream-ordered on the comm stream, nothing to wait on but the stream
                if m > 0:
                    self.comm.reduce_scatter_(self.flat[lo:lo + m]); self.launched_ops.append(("reduce_scatter", lo, lo + m))
                    self.comm.all_gather_(self.flat[lo:lo + m]); self.launched_ops.append(("all_gather", lo, lo + m))
                if lo + m < hi:
                    self.comm.all_reduce_(self.flat[lo + m:hi]); self.launched_ops.append(("all_reduce", lo + m, hi))
                return
            if m > 0 and self.flat.is_cuda and not self._stream_ordered():
                # gloo with a device arena (the shared-GPU plumbing tests; gloo moves device tensors through the host anyway and has no device
                # reduce-scatter): the same two collectives on a host copy of the slice, blocking
                host = self.flat[lo:lo + m].cpu()
                shard = host[self.rank * (m // W):(self.rank + 1) * (m // W)]
                dist.reduce_scatter_tensor(shard, host, op=dist.ReduceOp.SUM, group=self.pg)
                self.launched_ops.append(("reduce_scatter", lo, lo + m))
                dist.all_gather_into_tensor(host, shard.clone(), group=self.pg)
                self.launched_ops.append(("all_gather", lo, lo + m))
                self.flat[lo:lo + m].copy_(host)
            elif m > 0:
                seg = self.flat[lo:lo + m]
                shard = seg[self.rank * (m // W):(self.rank + 1) * (m // W)]
                w = dist.reduce_scatter_tensor(shard, seg, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                self.launched_ops.append(("reduce_scatter", lo, lo + m))
                if not self._stream_ordered():
                    w.wait()                                          # gloo: the gather below must see the reduced shard
                else:
                    self._works.append(w)
                self._works.append(dist.all_gather_into_tensor(seg, shard, group=self.pg, async_op=True))
                self.launched_ops.append(("all_gather", lo, lo + m))
            if lo + m < hi:                                            # all-reduce form, and the < world_size tail of an rs_ag slice
                self._works.append(dist.all_reduce(self.flat[lo + m:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                self.launched_ops.append(("all_reduce", lo + m, hi))

    def ready(self, lo: int, hi: int):
        """the backward finished gradient elements [lo, hi) (any order).  A range adjacent to a pending region merges into it; a range adjacent to none opens a
        new one — several regions may be pending at once (the SD3 full fine-tune walks its blocks AND the rows of its fused modulation matrix back to front, two
        interleaved descending sequences); a region goes out as soon as it holds a bucket's worth.  The slices and their order depend only on the sequence of
        ready() calls, which is the same on every replica."""
        if hi <= lo:
            return
        for r in self._pending:
            if hi == r[0]:
                r[0] = lo
                break
            if lo == r[1]:
                r[1] = hi
                break
        else:
            r = [lo, hi]
            self._pending.append(r)
        for q in self._pending:                      # a range that closed the gap between two pending regions joins them
            if q is not r and (q[1] == r[0] or r[1] == q[0]):
                r[0], r[1] = min(r[0], q[0]), max(r[1], q[1])
                self._pending.remove(q)
                break
        if r[1] - r[0] >= self.bucket_elems:
            self._pending.remove(r)
            self._fire(r[0], r[1])

    def all_reduce_now(self, flat: torch.Tensor):
        """blocking SUM of an arbitrary flat tensor over the group (the boundary step of a gradient accumulation: training.ddp_seam)"""
        if self.world_size <= 1:
            return
        if self.comm is not None:
            self.comm.all_reduce_(flat)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)

    def overlap_report(self) -> Optional[dict]:
        """ST355_COMM_TIMING=1: timestamps (ms from the start of the backward) of every slice's exchange on the comm stream, the end of the backward on the
        compute stream, the comm time left exposed after it and the overlapped fraction of the total comm time.  Synchronises; call between steps."""
        if not self.timing or self._ev_begin is None or self._ev_end is None:
            return None
        torch.cuda.synchronize(self.flat.device)
        t_bwd = self._ev_begin.elapsed_time(self._ev_end)
        sl = [dict(lo=lo, hi=hi, start_ms=self._ev_begin.elapsed_time(e0), end_ms=self._ev_begin.elapsed_time(e1)) for lo, hi, e0, e1 in self._ev_slices]
        comm = sum(s_["end_ms"] - s_["start_ms"] for s_ in sl)
        exposed = max(0.0, (max((s_["end_ms"] for s_ in sl), default=0.0)) - t_bwd)
        return dict(backward_ms=t_bwd, comm_ms=comm, exposed_ms=exposed, overlap_frac=max(0.0, 1.0 - exposed / comm) if comm > 0 else 1.0, slices=sl)

    def finish(self) -> float:
        """flush, join the comm stream back into the compute stream (device-side for nccl) and return the factor the optimizer must fold
        in (1/world_size)"""
        for r in self._pending:
            self._fire(r[0], r[1])
        self._pending = []
        if self.timing and self._ev_begin is not None:
            self._ev_end = torch.cuda.Event(enable_timing=True)
            self._ev_end.record(torch.cuda.current_stream(self.flat.device))     # the last backward kernel has been enqueued: everything after is exposed comm
        if self.comm_stream is not None and self._comm_used:
            with torch.cuda.stream(self.comm_stream):
                for w in self._works:
                    w.wait()
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm_stream)       # the optimizer (compute stream) starts after the last collective
        else:
            for w in self._works:
                w.wait()
        self._works, self._comm_used = [], False
        return 1.0 / self.world_size if self.enabled else 1.0

    @contextlib.contextmanager
    def no_sync(self):
        old = self.enabled
        self.enabled = False
        try:
            yield
        finally:
            self.enabled = old


def sync_module_states(module: torch.nn.Module, src: int = 0, process_group=None, chunk_bytes: int = 256 << 20) -> int:
    """What DDP does once at construction (torch's `_sync_module_states`, reached from accelerator.prepare, trainer.py:4515): every replica
    starts from rank `src`'s parameters and buffers.  Parameters here are views into a few large arenas (weights, K-extended LoRA columns,
    optimizer-ordered trainables), so the broadcast walks the distinct underlying STORAGES as raw bytes — a handful of GB-sized
    broadcasts over xGMI instead of thousands of per-tensor ones — and therefore also carries arena padding / fused columns that are not
    registered parameters.  Returns the number of bytes broadcast (0 when not distributed).  256 MiB per broadcast: the same bound GradSync keeps on every
    collective since RCCL's point-to-point path was seen delivering half of a > 1 GiB chunk (GradSync.max_slice_elems)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return 0
    seen, total = set(), 0
    tensors = list(module.parameters()) + list(module.buffers())
    for t in tensors:
        st = t.untyped_storage()
        key = st.data_ptr()
        if key in seen or st.nbytes() == 0:
            continue
        seen.add(key)
        raw = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st)        # the whole allocation, bytes
        for lo in range(0, raw.numel(), chunk_bytes):
            dist.broadcast(raw[lo:lo + chunk_bytes], src=src, group=process_group)
        total += raw.numel()
    refresh = getattr(module, "_refresh_transposed", None)                        # derived copies (transposed weights for dgrad) follow the new values
    if callable(refresh):
        refresh()
    for m in module.modules():                                                    # lazily built K-major copies (`prepare_for_training`) are rebuilt at the next forward
        if getattr(m, "_prepared", False):
            m._prepared = False
    return total
