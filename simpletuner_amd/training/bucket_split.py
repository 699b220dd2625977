"""Aspect buckets across data-parallel replicas (VERDICT r1 missing item 5).

Two pieces:

  * `split_buckets_between_processes` — the arithmetic of the reference's `MetadataBackend.split_buckets_between_processes`
    (helpers/metadata/backends/base.py:741-937): every bucket's sample list is canonically ordered, shuffled with the RUN seed (identical on every
    rank: `random.Random(f"{seed}:{id}:{bucket}")`), passed through the reference's trim (which never removes a sample: its bound
    ceil(total / effective) * effective is >= len(images) always — restated as written) and cut into contiguous per-rank slices (remainder to the low ranks; optional padding to equal length); buckets that cannot fill one effective batch raise the reference's
    ValueError, or — with oversubscription — are cycled up to a whole number of effective batches.  Pinned to the reference method executed here
    (tools/gen_bucket_golden.py -> tests/golden/bucket_split_vectors.pt).

  * `TokenBalancedSchedule` — what the reference does NOT do.  There every rank draws its next bucket independently
    (multiaspect/sampler.py:1041-1146), so in a step of an N-GPU job the ranks run different token counts and the gradient all-reduce waits
    for the slowest: with the 1024^2-area buckets of BASELINE configs[3] S ranges over 4032..4096 image tokens, with mixed-resolution datasets
    over far more.  Here all ranks walk ONE shared, seeded schedule of buckets — step t uses the same bucket everywhere, each rank takes the next
    micro-batch of its OWN slice of that bucket — so every step is token-balanced by construction, needs no communication, and the all-reduce
    never waits on a straggler.  The per-rank slices are the ones of the split above, so an epoch visits exactly the samples the reference's would.
"""
from __future__ import annotations

import random
from math import ceil
from typing import Dict, List, Optional, Sequence


def split_buckets_between_processes(buckets: Dict[str, Sequence], batch_size: int, num_processes: int, rank: int, gradient_accumulation_steps: int = 1,
                                    repeats: int = 0, seed: Optional[int] = 0, backend_id: str = "", shuffle: bool = True, apply_padding: bool = False,
                                    allow_oversubscription: bool = False, user_set_repeats: bool = False) -> Dict[str, List]:
    """helpers/metadata/backends/base.py:741-937 for a TRAINING dataset (eval datasets use an effective batch of 1 there and are outside the step path)"""
    effective = batch_size * num_processes * gradient_accumulation_steps
    failing = {b: len(v) for b, v in buckets.items() if v and len(v) * (repeats + 1) < effective}
    auto_repeats: Dict[str, int] = {}
    if failing:
        needed = {b: ceil(effective / n) - 1 for b, n in failing.items()}
        if allow_oversubscription and not user_set_repeats:
            auto_repeats = needed                      # pad only the undersized buckets (metadata/backends/base.py:813-823)
        else:
            lines = "".join(f"  - Bucket {b}: {n} samples x {repeats + 1} (with repeats) = {n * (repeats + 1)} samples; minimum repeats required: {needed[b]}\n"
                            for b, n in failing.items())
            raise ValueError("Dataset configuration will produce zero usable batches.\n"
                             f"  - Repeats: {repeats}\n  - Batch size: {batch_size}\n  - Number of GPUs: {num_processes}\n"
                             f"  - Gradient accumulation steps: {gradient_accumulation_steps}\n  - Effective batch size: {effective}\n"
                             f"Problem: {len(failing)} bucket(s) have insufficient samples:\n{lines}")
    out: Dict[str, List] = {}
    for bucket, images in buckets.items():
        if not images:
            out[bucket] = []
            continue
        images = list(images)
        if shuffle:
            images = sorted(images, key=str)           # canonical order first: every rank shuffles an identical sequence (metadata/backends/base.py:884-887)
            random.Random(f"{seed}:{backend_id}:{bucket}").shuffle(images)
        if bucket in auto_repeats:
            logical = len(images) * (auto_repeats[bucket] + 1)
            scheduled = ceil(logical / effective) * effective
            local = scheduled // num_processes
            start = rank * local
            out[bucket] = [images[(start + off) % len(images)] for off in range(local)]
            continue
        total = len(images) * (repeats + 1)
        trim = ceil(total / effective) * effective
        trimmed = images[:trim] if trim < len(images) else images
        per, extra = divmod(len(trimmed), num_processes)
        start = rank * per + min(rank, extra)
        size = per + int(rank < extra)
        part = trimmed[start:start + size]
        if apply_padding:
            target = per + int(extra > 0)
            if trimmed and len(part) < target:
                part = part + [trimmed[-1]] * (target - len(part))
        out[bucket] = part
    return out


class TokenBalancedSchedule:
    """one shared bucket order for all ranks (see module docstring).  `local_buckets` = this rank's split; `micro_batches_per_bucket` must be the SAME
    on every rank: with `counts=None` it is taken as the min over ranks of len(local bucket) // batch_size (`shared_counts`: one all-reduce when a
    multi-rank process group is initialised, the local value otherwise).  Iteration yields (bucket, [samples of this rank's micro-batch])."""

    def __init__(self, local_buckets: Dict[str, Sequence], batch_size: int, seed: int = 0, epoch: int = 0, counts: Optional[Dict[str, int]] = None,
                 tokens_of: Optional[Dict[str, int]] = None):
        self.local = {b: list(v) for b, v in local_buckets.items()}
        self.batch_size = int(batch_size)
        if counts is None:
            # under an initialised multi-rank process group the per-rank slices differ by one sample whenever len % world != 0 (the reference's trim is a
            # no-op: trim >= len(images) always, helpers/metadata/backends/base.py:760-763), so counts derived from THIS rank's slice would make the ranks
            # walk different bucket sequences: take the min over ranks (one tiny all-reduce at start-up) instead of silently diverging
            counts = shared_counts(self.local, self.batch_size)
        self.counts = dict(counts)
        for b, n in self.counts.items():
            if n * self.batch_size > len(self.local.get(b, ())):
                raise ValueError(f"bucket {b}: schedule asks for {n} micro-batches of {batch_size}, this rank holds {len(self.local.get(b, ()))} samples")
        self.tokens_of = dict(tokens_of or {})
        order = [b for b in sorted(self.counts, key=str) for _ in range(self.counts[b])]
        random.Random(f"st355-bucket-schedule:{seed}:{epoch}").shuffle(order)          # the SAME draw on every rank: seed and epoch are run-wide
        self.order = order

    def __len__(self):
        return len(self.order)

    def __iter__(self):
        cursor = {b: 0 for b in self.counts}
        for b in self.order:
            i = cursor[b]
            cursor[b] = i + 1
            yield b, self.local[b][i * self.batch_size:(i + 1) * self.batch_size]

    def step_tokens(self) -> List[int]:
        """tokens per step (identical on every rank by construction) — what the all-reduce balance argument rests on"""
        return [int(self.tokens_of.get(b, 0)) for b in self.order]


def shared_counts(local_buckets: Dict[str, Sequence], batch_size: int, process_group=None) -> Dict[str, int]:
    """micro-batches per bucket every rank can serve = min over ranks of len(local bucket) // batch_size (one tiny all-reduce at start-up)"""
    import torch
    import torch.distributed as dist
    names = sorted(local_buckets, key=str)
    mine = torch.tensor([len(local_buckets[b]) // batch_size for b in names], dtype=torch.int64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if str(dist.get_backend(process_group)).lower() == "nccl" else torch.device("cpu")
        t = mine.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=process_group)
        mine = t.cpu()
    return {b: int(n) for b, n in zip(names, mine.tolist())}
