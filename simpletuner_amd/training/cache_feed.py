"""DirectCacheFeeder — cache files -> pinned slabs -> HBM with ONE host copy per byte.

The reference's feed is `BatchFetcher` (data_backend/runtime/batch_fetcher.py:52-66) -> `collate_fn` (collate.py:590-1350) -> `prepare_batch`'s blocking
`.to(device)`: every latent / text-embed file is unpickled into a fresh pageable tensor (copy 1: file -> heap), stacked into the batch tensor (copy 2),
and copied to the device from pageable memory (copy 3 into the driver's bounce buffer, then the DMA).  At MI355X step rates (an 8-GPU node consumes
~45 Flux images/s = 200 MB/s of latents + embeddings, the SDXL-LoRA config ~190 images/s) the host side must not become the critical path.

Here, per batch:
  * every cache file is opened with `torch.load(path, mmap=True, weights_only=True)` (local, uncompressed files: the tensor aliases the page cache —
    no read copy; gzip-wrapped / remote payloads fall back to `loads_cache_payload`, cache_io.py);
  * a pool of worker threads copies each sample STRAIGHT into its row of a pinned slab `[B, ...]` (the only host copy: page cache -> pinned), running
    the reference's per-latent sanity checks (shape agreement, finite values, collate.py:526-587) on the way;
  * the slabs go to the device with asynchronous copies on a dedicated copy stream; `next()` makes the training stream wait on the batch's event
    (device-side) and hands out the same keys `collate.assemble_batch` produces for the plain case (latent_batch, prompt_embeds, add_text_embeds,
    encoder_attention_mask, filepaths, prompts, data_backend_id).
`slots` batches are in flight (slab reuse is fenced by the previous copy's event), so file reads, host copies and DMA of batch i+1.. overlap the
train step of batch i.  Caption dropout / conditioning inputs keep going through `assemble_batch` + `PinnedBatchStager` (collate.py) — same
results, two more host copies.  Throughput evidence: tools/cache_feed_bench.py -> profiles/archive/r02_cache_feed.json.
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence

import torch

from .cache_io import GZIP_MAGIC, CacheReader, latent_from_payload, loads_cache_payload

_TEXT_FIELDS = (("prompt_embeds", "prompt_embeds"), ("pooled_prompt_embeds", "add_text_embeds"), ("attention_masks", "encoder_attention_mask"),
                ("attention_mask", "encoder_attention_mask"), ("prompt_attention_mask", "encoder_attention_mask"))


def load_cache_file(path: str, read_bytes=None):
    """one cache file -> the saved object; local uncompressed files are memory-mapped (tensors alias the page cache)"""
    if read_bytes is None:
        try:
            with open(path, "rb") as fh:
                head = fh.read(2)
        except FileNotFoundError:
            raise FileNotFoundError(f"{path} not found.")
        if head != GZIP_MAGIC:
            return torch.load(path, map_location="cpu", mmap=True, weights_only=True)
        with open(path, "rb") as fh:
            return loads_cache_payload(fh.read())
    return loads_cache_payload(read_bytes(path))


class DirectCacheFeeder:
    def __init__(self, reader: CacheReader, device, dtype=torch.bfloat16, slots: int = 3, workers: int = 16, depth: int = 2):
        if not torch.cuda.is_available():
            raise RuntimeError("DirectCacheFeeder needs the GPU runtime (pinned host memory + a HIP copy stream); there is no CPU path")
        self.reader, self.device, self.dtype = reader, torch.device(device), dtype
        self.stream = torch.cuda.Stream(device=self.device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)), thread_name_prefix="st355-cache")
        self.slots = [dict(pinned={}, done=None) for _ in range(max(2, int(slots)))]
        self._next_slot = 0
        self._todo: "queue.Queue" = queue.Queue()
        self._ready: "queue.Queue" = queue.Queue(maxsize=max(1, int(depth)))
        self._err: Optional[BaseException] = None
        from . import cache_io as _cio
        self._custom_read = None if reader.read_bytes is _cio._read_local else reader.read_bytes     # identity, not __name__: partials / callable objects are custom readers
        self._t = threading.Thread(target=self._run, name="st355-cache-feed", daemon=True)
        self._t.start()

    # ---- producer side ----
    def submit(self, examples: Sequence[dict]) -> None:
        """queue one batch (the sampler's example records: image_path, instance_prompt_text, ...)"""
        self._todo.put(list(examples))

    def close(self) -> None:
        self._todo.put(None)

    def _slab(self, slot, key, shape, dtype):
        buf = slot["pinned"].get(key)
        if buf is None or tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
            buf = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
            slot["pinned"][key] = buf
        return buf

    def _sample(self, slot, i, B, ex, shapes, lock, counts):
        path = self.reader.latent_path(ex["image_path"])
        lat, _meta = latent_from_payload(load_cache_file(path, self._custom_read))
        if lat.dim() == 5:
            lat = lat[0]
        if not torch.isfinite(lat).all():                                   # collate.py:560-566 (the cache owner deletes the file; here: refuse)
            raise ValueError(f"(id={ex.get('data_backend_id')}) cache file {path}: contains NaN or Inf values")
        rec = load_cache_file(self.reader.text_path(ex.get("instance_prompt_text") or ""), self._custom_read)
        if not isinstance(rec, dict):
            raise ValueError(f"text-embed cache entry is a {type(rec).__name__}; this path reads the dict layout")
        fields = {"latent_batch": lat}
        for src, dst in _TEXT_FIELDS:
            if src in rec and dst not in fields:
                t = rec[src]
                if t.dim() == 3 and t.shape[0] != 1:
                    raise ValueError(f"text-embed cache entry {src} holds {t.shape[0]} rows; one example per record expected")
                fields[dst] = t[0] if t.dim() == 3 else t          # collate_tensors (collate.py:409-451): [1,S,D] records are concatenated, [S,D] / [D] stacked
        with lock:
            for key in fields:
                counts[key] = counts.get(key, 0) + 1
        for key, t in fields.items():
            with lock:
                want = shapes.setdefault(key, tuple(t.shape))
            if tuple(t.shape) != want:
                raise ValueError(f"(id={ex.get('data_backend_id')}) File {path} {key} shape mismatch: {tuple(t.shape)} != {want}")
            dt = self.dtype if t.is_floating_point() else t.dtype
            self._slab_locked(slot, key, (B,) + want, dt, lock)[i].copy_(t)          # page cache -> pinned: the one host copy

    def _slab_locked(self, slot, key, shape, dtype, lock):
        with lock:
            return self._slab(slot, key, shape, dtype)

    def _run(self):
        try:
            while True:
                examples = self._todo.get()
                if examples is None:
                    self._ready.put(None)
                    return
                slot = self.slots[self._next_slot]
                self._next_slot = (self._next_slot + 1) % len(self.slots)
                if slot["done"] is not None:
                    slot["done"].synchronize()                               # the slab's previous DMA has left the host
                B, shapes, lock, counts = len(examples), {}, threading.Lock(), {}
                self._sample(slot, 0, B, examples[0], shapes, lock, counts)   # the first sample fixes the shapes (slab allocation)
                list(self.pool.map(lambda ie: self._sample(slot, ie[0], B, ie[1], shapes, lock, counts), list(enumerate(examples))[1:]))
                short = {k: c for k, c in counts.items() if c != B}
                if short:       # the slabs are reused: a row no sample wrote would carry the PREVIOUS batch's data into this one
                    raise ValueError(f"batch fields not present in every sample (rows written of {B}): {short}")
                ars = {ex.get("aspect_ratio") for ex in examples if ex.get("aspect_ratio") is not None}
                if len(ars) > 1:
                    raise ValueError(f"Aspect ratio mismatch inside one batch: {sorted(ars)}")
                out: Dict[str, object] = {}
                with torch.cuda.stream(self.stream):
                    for key in shapes:
                        out[key] = slot["pinned"][key].to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                slot["done"] = ev
                out.update(filepaths=[ex.get("image_path") for ex in examples], prompts=[ex.get("instance_prompt_text") for ex in examples],
                           data_backend_id=examples[-1].get("data_backend_id"), _ready=ev)
                self._ready.put(out)
        except BaseException as e:                                            # surfaced on the training thread
            self._err = e
            self._ready.put(None)

    # ---- consumer side (the training thread) ----
    def next(self) -> Optional[dict]:
        b = self._ready.get()
        if self._err is not None:
            raise self._err
        if b is None:
            return None
        ev = b.pop("_ready")
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)                                                    # device-side: the host never blocks on the copy
        for v in b.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return b
