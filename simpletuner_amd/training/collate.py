"""Batch assembly for the step loop — SURVEY.md §8(a) row 2: the tensor work of the reference's `collate_fn`
(simpletuner/helpers/training/collate.py:590-1350; keys of the returned dict :1316-1350) and its helpers (`compute_time_ids` :59-98,
`compute_prompt_embeddings._collate_tensors` :409-451 and the key mapping :454-481, `gather_conditional_*_size_features` :487-523,
`check_latent_shapes` :526-587), plus the MI355X side of the hand-over the reference leaves to `.to(device)` calls inside `prepare_batch`.

What is and is not here.  The reference's collate_fn is 90 % control plane — StateTracker look-ups, cache back-ends, path mapping, dataset
types this tier does not cover (audio, video, grounding, DeepFloyd pixels).  None of that is rebuilt.  What the hot path needs is the part that
turns per-example cache records (a latent tensor + the text-encoder outputs read from the caches) into ONE batch dict of stacked tensors with
the reference's key names — `assemble_batch` — and a way to get those host tensors into HBM without stalling the step: `PinnedBatchStager`.

MI355X form of the hand-over: the reference stacks on the host, then copies every field with a blocking `.to(device)` from pageable memory at
the top of `prepare_batch` (common.py:5872-5935) — per step, on the compute stream.  Here every tensor field of the assembled batch is copied
once into a pinned staging slab that is allocated on first use and then reused (no per-step page-locking; `check_latent_shapes(out=slab)` can
stack the latents straight into it), and the slab is sent with one asynchronous copy per field on a dedicated copy stream; the step only waits
on an event, on the device.  Two slabs alternate, so batch i+1 is staged while step i computes (the reference's
`BatchFetcher` thread, batch_fetcher.py:52-66, overlaps only the host part).  A Flux batch of 8 is 8x(16x128x128 + 512x4096 + 768) bf16
= 37.8 MB: ~0.7 ms over PCIe gen5 x16, hidden entirely.
"""
from __future__ import annotations

import threading
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch

# the keys collate_fn returns (collate.py:1316-1350) that an image-model step reads; everything else in that dict belongs to dataset types
# outside this tier and is passed through untouched when the caller supplies it
TEXT_KEY_MAP = (("prompt_embeds", "prompt_embeds"), ("pooled_prompt_embeds", "add_text_embeds"), ("attention_masks", "encoder_attention_mask"),
                ("batch_time_ids", "batch_time_ids"))


def compute_time_ids(intermediary_size, target_size, weight_dtype, vae_downscale_factor: int = 8, crop_coordinates=None,
                     refiner_aesthetic_score: Optional[float] = None) -> torch.Tensor:
    """SDXL micro-conditioning row [[orig_h, orig_w, crop_top, crop_left, target_h, target_w]] (collate.py:59-98).  `intermediary_size` is
    (width, height) of the image after the resize that precedes the crop; `target_size` is the LATENT shape (C, H, W).  The refiner replaces the
    target size by its aesthetic score (:91-93) — passed explicitly instead of read from global state."""
    if intermediary_size is None or target_size is None:
        raise Exception(f"Cannot continue, the intermediary_size or target_size were not provided: {intermediary_size}, {target_size}")
    width, height = intermediary_size[0], intermediary_size[1]
    if width is None:
        raise ValueError("Original width must be specified.")
    if height is None:
        raise ValueError("Original height must be specified.")
    if crop_coordinates is None:
        raise ValueError("Crop coordinates were not collected during collate.")
    tail = ((refiner_aesthetic_score,) if refiner_aesthetic_score is not None
            else (int(target_size[1] * vae_downscale_factor), int(target_size[2] * vae_downscale_factor)))
    return torch.tensor([list((height, width) + tuple(crop_coordinates) + tail)], dtype=weight_dtype)


def gather_conditional_sdxl_size_features(examples: Sequence[dict], latents, weight_dtype) -> torch.Tensor:
    """[B, 1, 6] time ids, zeroed for examples whose conditioning was dropped (collate.py:501-523)"""
    if len(examples) != len(latents):
        raise ValueError(f"Number of examples ({len(examples)}) and latents ({len(latents)}) must match.")
    rows = []
    for ex, lat in zip(examples, latents):
        row = compute_time_ids(tuple(ex.get("intermediary_size", ex.get("original_size"))), lat.shape, weight_dtype,
                               crop_coordinates=ex["crop_coordinates"])
        rows.append(torch.zeros_like(row) if ex["drop_conditioning"] else row)
    return torch.stack(rows, dim=0)


def gather_conditional_pixart_size_features(examples: Sequence, latents: torch.Tensor, weight_dtype, device=None) -> Dict[str, torch.Tensor]:
    """PixArt's `added_cond_kwargs` (collate.py:487-498): resolution [B, 2] = pixel (H, W) of the bucket, aspect_ratio [B, 1] = H / W"""
    bsz = len(examples)
    h, w = latents.shape[2] * 8, latents.shape[3] * 8
    return {"resolution": torch.tensor([h, w]).repeat(bsz, 1).to(dtype=weight_dtype, device=device),
            "aspect_ratio": torch.tensor([float(h / w)]).repeat(bsz, 1).to(dtype=weight_dtype, device=device)}


def collate_tensors(tensors: Iterable[Optional[torch.Tensor]]) -> Optional[torch.Tensor]:
    """per-example cache entries -> one batch tensor (collate.py:409-451): [S,D] and [D] entries are stacked, [1,S,D] (or [b,S,D]) entries are
    concatenated on the batch axis, a mix is first brought to 3-D"""
    ts = [t for t in tensors if t is not None]
    if not ts:
        return None
    nd = ts[0].dim()
    if nd in (1, 2):
        return torch.stack(ts)
    if nd == 3 and all(t.dim() == 3 for t in ts):
        return torch.cat(ts, dim=0)
    lifted = []
    for t in ts:
        if t.dim() in (1, 2):
            lifted.append(t.unsqueeze(0))
        elif t.dim() == 3:
            lifted.append(t)
        else:
            raise ValueError(f"Unexpected tensor dimension: {t.dim()} with shape {t.shape}")
    return torch.cat(lifted, dim=0)


def collate_prompt_embeds(text_encoder_output: Sequence[dict], model=None) -> Dict[str, torch.Tensor]:
    """collate.py:453-485: the model may collate its own way (`model.collate_prompt_embeds`, common.py:1768-1774; {} = no opinion); otherwise every
    known field of the per-example records is collated.  Old- and new-style attention mask names all land in `attention_masks`."""
    out = dict(model.collate_prompt_embeds(text_encoder_output)) if model is not None and hasattr(model, "collate_prompt_embeds") else {}
    if not out:
        first = text_encoder_output[0]
        for src, dst in (("prompt_embeds", "prompt_embeds"), ("pooled_prompt_embeds", "pooled_prompt_embeds"), ("attention_mask", "attention_masks"),
                         ("prompt_attention_mask", "attention_masks"), ("attention_masks", "attention_masks"), ("time_ids", "time_ids")):
            if src in first:
                out[dst] = collate_tensors([rec[src] for rec in text_encoder_output])
    if not out:
        raise Exception(f"Could not compute text encoder output: {text_encoder_output}")
    return out


def check_latent_shapes(latents: Sequence[Optional[torch.Tensor]], filepaths: Sequence[str], data_backend_id, batch: Sequence, is_conditioning: bool = False,
                        on_corrupt: Optional[Callable[[str], None]] = None, out: Optional[torch.Tensor] = None):
    """collate.py:526-587: one aspect ratio per training batch, no missing / non-finite latents (the reference deletes the offending cache file —
    here `on_corrupt(filepath)` is called so the cache owner can), identical shapes -> stacked [B, ...]; conditioning latents of differing shapes
    are returned as a list.  `out`: stack straight into this (pinned) tensor instead of allocating."""
    want = latents[0].shape if latents[0] is not None else None
    if want is not None and len(want) == 5:
        want = want[1:]
    if not is_conditioning:
        first_ar = None
        for ex in batch:
            ar = ex.get("aspect_ratio") if isinstance(ex, dict) else getattr(ex, "aspect_ratio", None)
            if first_ar is None and ar is not None:
                first_ar = ar
            if ar is not None and first_ar is not None and ar != first_ar:
                raise ValueError(f"(id=({data_backend_id}) Aspect ratio mismatch: {ar} != {first_ar}")
    for i, lat in enumerate(latents):
        if lat is None:
            raise ValueError(f"(id={data_backend_id}) File {filepaths[i]} latent is None.")
        if not torch.isfinite(lat).all():
            if on_corrupt is not None:
                on_corrupt(filepaths[i])
            raise ValueError(f"(id={data_backend_id}) Deleted cache file {filepaths[i]}: contains NaN or Inf values")
        if not is_conditioning:
            got = lat.shape[1:] if len(lat.shape) == 5 else lat.shape
            if got != want:
                raise ValueError(f"(id={data_backend_id}) File {filepaths[i]} latent shape mismatch: {got} != {want}")
    if is_conditioning and len({tuple(l.shape) for l in latents}) > 1:
        return list(latents)
    if out is not None:
        for i, lat in enumerate(latents):
            out[i].copy_(lat)
        return out
    return torch.stack(list(latents), dim=0)


def assemble_batch(examples: Sequence[dict], latents: Sequence[torch.Tensor], text_encoder_output: Sequence[dict], *, model=None, model_family: Optional[str] = None,
                   weight_dtype=torch.bfloat16,
                   data_backend_id: Optional[str] = None, conditioning_latents: Optional[Sequence[torch.Tensor]] = None,
                   caption_dropout_probability: Optional[float] = None, empty_prompt_record: Optional[dict] = None, draw: Callable[[], float] = None,
                   extra: Optional[dict] = None) -> dict:
    """The tensor part of collate_fn for an image-model step: per-example cache records -> the batch dict `prepare_batch` consumes.

    examples: the sampler's records (`image_path`, `instance_prompt_text`, `crop_coordinates`, `intermediary_size` / `original_size`, `aspect_ratio`,
    `data_backend_id`); latents / text_encoder_output: what the VAE / text-embed caches returned for them, in order.  Caption dropout
    (collate.py:613-621): with probability p an example's caption is emptied and `drop_conditioning` set — its text record is replaced by
    `empty_prompt_record` (the cached embedding of "") and, for SDXL, its time ids are zeroed."""
    if len(examples) != len(latents) or len(examples) != len(text_encoder_output):
        raise ValueError(f"Number of examples ({len(examples)}), latents ({len(latents)}) and text records ({len(text_encoder_output)}) must match.")
    if draw is None:
        import random
        draw = random.random
    text_encoder_output = list(text_encoder_output)
    for i, ex in enumerate(examples):
        dropped = bool(caption_dropout_probability) and caption_dropout_probability > 0 and draw() < caption_dropout_probability
        ex["drop_conditioning"] = dropped
        if dropped:
            ex["instance_prompt_text"] = ""
            if empty_prompt_record is None:
                raise ValueError("caption dropout needs the cached embedding of the empty prompt (empty_prompt_record)")
            text_encoder_output[i] = empty_prompt_record
    backend = data_backend_id if data_backend_id is not None else (examples[-1].get("data_backend_id") if examples else None)
    filepaths = [ex.get("image_path") for ex in examples]
    latent_batch = check_latent_shapes(latents, filepaths, backend, examples)
    text = collate_prompt_embeds(text_encoder_output, model)
    if model_family in ("sdxl", "kolors"):                                   # collate.py:1148-1160 (model-specific logic the reference keeps in collate)
        text["batch_time_ids"] = gather_conditional_sdxl_size_features(examples, latent_batch, weight_dtype)
    elif model_family == "pixart_sigma":
        text["batch_time_ids"] = gather_conditional_pixart_size_features(examples, latent_batch, weight_dtype)
    batch = {"latent_batch": latent_batch, "latent_metadata": None, "filepaths": filepaths, "data_backend_id": backend,
             "prompts": [ex.get("instance_prompt_text") for ex in examples], "text_encoder_output": text,
             "conditioning_latents": None, "conditioning_pixel_values": None, "conditioning_type": None, "loss_mask_type": None,
             "is_regularisation_data": False, "is_i2v_data": False}
    for src, dst in TEXT_KEY_MAP:
        batch[dst] = text.get(src)
    if conditioning_latents is not None:
        if len(conditioning_latents) != len(examples):
            raise ValueError("Number of training samples and conditioning samples must match for ControlNet.")
        batch["conditioning_latents"] = check_latent_shapes(conditioning_latents, filepaths, backend, examples, is_conditioning=True)
    if extra:
        batch.update(extra)
    return batch


class PinnedBatchStager:
    """Host batch dict -> HBM through pinned slabs and a copy stream (see the module docstring).  `stage()` returns a dict whose tensors are
    device tensors plus `_ready`, an event the consumer's stream must wait on (`wait()` does it); everything that is not a tensor is passed
    through.  `slots` slabs rotate: the slab of batch i is reused for batch i+slots, after its copy has completed."""

    def __init__(self, device, slots: int = 2):
        if not torch.cuda.is_available():
            raise RuntimeError("PinnedBatchStager needs the GPU runtime (pinned host memory + a HIP copy stream); there is no CPU path")
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.slots = [dict(pinned={}, done=None) for _ in range(max(2, int(slots)))]
        self._next = 0

    def _slab(self, slot, key, like: torch.Tensor) -> torch.Tensor:
        buf = slot["pinned"].get(key)
        if buf is None or buf.shape != like.shape or buf.dtype != like.dtype:
            buf = torch.empty(like.shape, dtype=like.dtype, pin_memory=True)
            slot["pinned"][key] = buf
        return buf

    def stage(self, batch: dict, dtype_map: Optional[Dict[str, torch.dtype]] = None) -> dict:
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        if slot["done"] is not None:
            slot["done"].synchronize()                       # the slab's previous copy has left the host
        out, todo = {}, []

        def visit(key, val):
            if torch.is_tensor(val) and val.device.type == "cpu":
                want = (dtype_map or {}).get(key.split("/")[-1])
                src = val if want is None or val.dtype == want or not val.is_floating_point() else val.to(want)
                pin = self._slab(slot, key, src)
                pin.copy_(src)                               # pageable -> pinned, one pass
                todo.append((key, pin))
                return None
            if isinstance(val, dict):
                return {k: visit(f"{key}/{k}", v) for k, v in val.items()}
            return val

        staged = {k: visit(k, v) for k, v in batch.items()}
        dev = {}
        with torch.cuda.stream(self.stream):
            for key, pin in todo:
                dev[key] = pin.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot["done"] = ev

        def fill(key, val):
            if val is None and key in dev:
                return dev[key]
            if isinstance(val, dict):
                return {k: fill(f"{key}/{k}", v) for k, v in val.items()}
            return val

        out = {k: fill(k, v) for k, v in staged.items()}
        out["_ready"] = ev
        return out

    @staticmethod
    def wait(batch: dict) -> dict:
        """make the CURRENT stream wait for the batch's copies (device-side wait, no host sync); drops the `_ready` key"""
        ev = batch.pop("_ready", None)
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)

            def mark(v):                                     # the tensors were allocated on the copy stream: tell the allocator who reads them
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
                elif isinstance(v, dict):
                    for x in v.values():
                        mark(x)
            mark(batch)
        return batch


class Prefetcher:
    """the reference's BatchFetcher (batch_fetcher.py:52-66: a thread that keeps a small queue of collated batches) with the staging step moved
    into the thread: `next()` hands back a batch whose H2D copies are already in flight."""

    def __init__(self, make_batch: Callable[[], Optional[dict]], stager: Optional[PinnedBatchStager] = None, depth: int = 2,
                 dtype_map: Optional[Dict[str, torch.dtype]] = None):
        import queue
        self._make, self._stager, self._dtype_map = make_batch, stager, dtype_map
        self._q = queue.Queue(maxsize=max(1, depth))
        self._stop = threading.Event()
        self._err = None
        self._t = threading.Thread(target=self._run, name="st355-prefetch", daemon=True)
        self._t.start()

    def _run(self):
        try:
            while not self._stop.is_set():
                b = self._make()
                if b and self._stager is not None:
                    b = self._stager.stage(b, self._dtype_map)
                self._q.put(b)
                if not b:                                    # falsy batch = the epoch-end sentinel (trainer.py:6974): deliver it and stop
                    return
        except BaseException as e:                           # surfaced on the training thread
            self._err = e
            self._q.put(None)

    def next(self):
        b = self._q.get()
        if self._err is not None:
            raise self._err
        if b and self._stager is not None:
            PinnedBatchStager.wait(b)
        return b

    def close(self):
        self._stop.set()
        while not self._q.empty():
            self._q.get_nowait()
