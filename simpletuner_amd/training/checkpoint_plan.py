"""Activation-checkpoint plans shared by every transformer family of the st355 path.

The reference decides, per `gradient_checkpointing` / `gradient_checkpointing_interval` / `gradient_checkpointing_segment_stride`, which blocks run
under its checkpoint function: `checkpoint_sequential_state` (helpers/training/gradient_checkpointing_interval.py:69-120) when the interval is > 1 —
call sites flux/transformer.py:1142-1209 and 1331-1363, sd3/transformer.py:716-752, pixart/transformer.py:627-672 — and one checkpoint per block
otherwise (`should_checkpoint_block`, :48-66).  The plans below reproduce the segments the reference was RECORDED to wrap when its own model files
were executed (tests/golden/ref_{flux,sd3}_model.pt, `checkpoint_plans`; tests/test_ref_models_cpu.py), mode for mode.

On this path "checkpointed" means: only the segment's input is kept; backward re-runs the segment's forward with the same kernels in the same order
(bit-identical activations, hence bit-identical gradients) and then walks it backwards."""
from __future__ import annotations

from typing import List, Optional, Tuple


def segments(n_blocks: int, enabled: bool, interval: Optional[int] = None, stride: Optional[int] = None) -> List[Tuple[int, int, bool]]:
    """[(first block, block count, recompute?)] over one block stack:
         checkpointing off                              -> every block keeps its activations;
         on, interval None / <= 1  ("layer")            -> every block is its own checkpoint;
         on, interval k > 1 [, segment_stride s >= k]   -> the first k blocks of every s-block window form ONE checkpoint, the s - k blocks of the
                                                           gap keep their activations."""
    if not enabled:
        return [(i, 1, False) for i in range(n_blocks)]
    k = interval
    if k is None or k <= 1:
        return [(i, 1, True) for i in range(n_blocks)]
    s = stride or k
    if s < k:
        raise ValueError("segment_stride must be at least segment_size")
    segs: List[Tuple[int, int, bool]] = []
    for s0 in range(0, n_blocks, s):
        n = min(k, n_blocks - s0)
        segs.append((s0, n, True))
        segs += [(j, 1, False) for j in range(s0 + n, min(s0 + s, n_blocks))]
    return segs


def per_block(n_blocks: int, enabled: bool, interval: Optional[int] = None, stride: Optional[int] = None) -> List[Tuple[int, int, bool]]:
    """one entry per block, checkpointed where `should_checkpoint_block` says so (gradient_checkpointing_interval.py:48-66): what the reference falls back to
    when the segmented form is off — under TREAD routing, ControlNet residuals, skipped layers (sd3/transformer.py:716-728, 768-777)"""
    out = []
    for i in range(n_blocks):
        if not enabled:
            ck = False
        elif interval is None or interval <= 1:
            ck = True
        elif stride is None:
            ck = i % interval == 0
        else:
            if stride < interval:
                raise ValueError("segment_stride must be at least interval")
            ck = i % stride < interval
        out.append((i, 1, ck))
    return out


def wrapped(n_blocks: int, interval: Optional[int] = None, stride: Optional[int] = None) -> List[List[int]]:
    """the block indices of every recomputed segment (what the reference hands to its checkpoint function)"""
    return [list(range(a, a + n)) for a, n, rc in segments(n_blocks, True, interval, stride) if rc]


flux_segments = wrapped      # per stack: the double and the single blocks are planned separately (flux/transformer.py:1170, 1334)
sd3_segments = wrapped
pixart_segments = wrapped


class CheckpointPlanMixin:
    """the gradient-checkpointing switches of the reference's transformer classes (flux/transformer.py:816-835, sd3/transformer.py:394-405,
    pixart/transformer.py:431-439): `gradient_checkpointing` + `set_gradient_checkpointing_interval / _segment_stride / _backend`.  The engine of the
    class that mixes this in asks `_checkpoint_segments(n_blocks)` for its plan."""
    gradient_checkpointing = False
    gradient_checkpointing_interval = None
    gradient_checkpointing_segment_stride = None
    gradient_checkpointing_backend = "torch"

    def set_gradient_checkpointing_interval(self, value):
        self.gradient_checkpointing_interval = None if value is None else int(value)

    def set_gradient_checkpointing_segment_stride(self, segment_stride):
        self.gradient_checkpointing_segment_stride = None if segment_stride is None else int(segment_stride)

    def set_gradient_checkpointing_backend(self, backend: str):
        if backend != "torch":      # "unsloth" = CPU-offloaded checkpoints (pointless with 288 GB), "*-ffn" = FFN-only scope: not built, never silently ignored
            raise NotImplementedError(f"gradient_checkpointing_backend={backend!r} is not implemented on the st355 path (built: 'torch' = recompute)")
        self.gradient_checkpointing_backend = backend

    def enable_gradient_checkpointing(self, *a, **k):
        self.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    def _checkpoint_segments(self, n_blocks: int):
        return segments(n_blocks, self.gradient_checkpointing, self.gradient_checkpointing_interval, self.gradient_checkpointing_segment_stride)
