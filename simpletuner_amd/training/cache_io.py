"""Reading the reference's on-disk latent / text-embed caches — SURVEY.md §8(f)1 ("keep the on-disk `.pt` layout readable").

Only the FORMAT side is here, so a cache directory written by stock SimpleTuner feeds `assemble_batch` directly:
  * file naming: VAE cache `<cache_dir>/[<sub-folders under instance_data_dir>/]<image stem | sha256(stem)>.pt`
    (caching/vae.py:678-703); text-embed cache `<cache_dir>/<md5(key)[ \\0prompt\\0 prompt]>-<model_type>.pt` (caching/text_embeds.py:126-182,
    caption-keyed and path-keyed variants);
  * payload: `torch.save` of a tensor (latents), of a dict holding `latents` (+ metadata), or of the text-encoder output dict / tuple — optionally
    gzip-wrapped when the backend was configured with `compress_cache` (data_backend/base.py:126-153, local.py:309-359).  Writes are atomic
    (temp file + rename), as the reference's `atomic_write`.
The storage back-ends themselves (S3, CSV, HF datasets, webdataset …) are control plane and stay out of scope: `CacheReader` takes any
`read_bytes(path) -> bytes` callable, the local file system by default.
"""
from __future__ import annotations

import gzip
import hashlib
import io
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

GZIP_MAGIC = b"\x1f\x8b"


def vae_cache_filename(filepath: str, cache_dir: str, instance_data_dir: Optional[str] = None, hash_filenames: bool = False) -> Tuple[str, str]:
    """(full path, base name) of the latent cache entry of an image path (caching/vae.py:678-703; sample-id back-ends excluded)"""
    if filepath.endswith(".pt"):
        return filepath, os.path.basename(filepath)
    stem = os.path.splitext(os.path.basename(filepath))[0]
    if hash_filenames:
        stem = hashlib.sha256(str(stem).encode()).hexdigest()
    base = str(stem) + ".pt"
    sub = ""
    if instance_data_dir is not None:
        sub = os.path.dirname(filepath).replace(instance_data_dir, "").lstrip(os.sep)
    return (os.path.join(cache_dir, sub, base) if sub else os.path.join(cache_dir, base)), base


def text_embed_cache_filename(key_value, cache_dir: str, model_type: str, prompt: Optional[str] = None, path_based_keys: bool = False,
                              filename_key: bool = False) -> str:
    """caching/text_embeds.py:126-182.  Caption-keyed caches hash the caption itself (key_value = the prompt); path-keyed caches (models whose
    embeddings depend on the sample) hash the normalised path and mix the prompt in after a `\\0prompt\\0` separator."""
    key = "" if key_value is None else str(key_value)
    if filename_key and "://" not in key:
        key = os.path.normcase(os.path.abspath(os.path.normpath(key)))
    h = hashlib.md5()
    h.update(key.encode())
    if path_based_keys and prompt:
        h.update(b"\0prompt\0")
        h.update(str(prompt).encode())
    return os.path.join(cache_dir, f"{h.hexdigest()}-{model_type}.pt")


def loads_cache_payload(raw: bytes, allow_pickle: bool = False):
    """bytes of one cache file -> the saved object on the CPU; transparently unwraps the gzip container of `compress_cache` back-ends.
    Loaded with `weights_only=True` — what the reference's `torch.load(..., map_location="cpu")` means on current torch — so a cache file from
    shared / remote storage cannot execute a pickle; `allow_pickle=True` is the explicit opt-in for legacy files that hold arbitrary objects."""
    if raw[:2] == GZIP_MAGIC:
        raw = gzip.decompress(raw)
    return torch.load(io.BytesIO(raw), map_location="cpu", weights_only=not allow_pickle)


def dumps_cache_payload(data, compress: bool = False) -> bytes:
    buf = io.BytesIO()
    torch.save(data, buf)
    return gzip.compress(buf.getvalue()) if compress else buf.getvalue()


def save_cache_file(path: str, data, compress: bool = False) -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as fh:
        fh.write(dumps_cache_payload(data, compress))
        fh.flush()
        os.fsync(fh.fileno())
    os.replace(tmp, path)


def _read_local(path: str) -> bytes:
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found.")
    with open(path, "rb") as fh:
        return fh.read()


def latent_from_payload(obj) -> Tuple[torch.Tensor, Optional[dict]]:
    """a cache entry is the latent tensor, or a dict with `latents` + metadata (caching/vae.py:1418-1441; collate.py:663-671)"""
    if isinstance(obj, dict):
        if "latents" not in obj:
            raise ValueError(f"latent cache entry is a dict without 'latents' (keys: {sorted(obj)})")
        return obj["latents"], {k: v for k, v in obj.items() if k != "latents"}
    if not torch.is_tensor(obj):
        raise ValueError(f"latent cache entry is a {type(obj).__name__}, expected a tensor or a dict with 'latents'")
    return obj, None


class CacheReader:
    """examples -> (latents, text-encoder records) for `collate.assemble_batch`, read from cache directories in the reference's layout"""

    def __init__(self, vae_cache_dir: str, text_cache_dir: str, model_type: str, instance_data_dir: Optional[str] = None, hash_filenames: bool = False,
                 read_bytes: Callable[[str], bytes] = _read_local, workers: int = 8):
        self.vae_cache_dir, self.text_cache_dir, self.model_type = vae_cache_dir, text_cache_dir, model_type
        self.instance_data_dir, self.hash_filenames = instance_data_dir, hash_filenames
        self.read_bytes, self.workers = read_bytes, max(1, int(workers))

    def latent_path(self, image_path: str) -> str:
        return vae_cache_filename(image_path, self.vae_cache_dir, self.instance_data_dir, self.hash_filenames)[0]

    def text_path(self, prompt: str) -> str:
        return text_embed_cache_filename(prompt, self.text_cache_dir, self.model_type)

    def read(self, examples: Sequence[dict]) -> Tuple[List[torch.Tensor], List[dict]]:
        """one thread per file up to `workers` (the reference reads text embeds through a ThreadPoolExecutor, collate.py:397-407, and latents through
        its VAE cache's own pool): the reads are independent small files, latency-bound"""
        from concurrent.futures import ThreadPoolExecutor
        jobs = [("l", self.latent_path(ex["image_path"])) for ex in examples] + [("t", self.text_path(ex.get("instance_prompt_text") or "")) for ex in examples]
        with ThreadPoolExecutor(max_workers=min(self.workers, len(jobs) or 1)) as pool:
            raw = list(pool.map(lambda j: loads_cache_payload(self.read_bytes(j[1])), jobs))
        n = len(examples)
        latents = [latent_from_payload(o)[0] for o in raw[:n]]
        records: List[Dict] = []
        for o in raw[n:]:
            if not isinstance(o, dict):
                raise ValueError(f"text-embed cache entry is a {type(o).__name__}; this path reads the dict layout (prompt_embeds / pooled_prompt_embeds / attention_masks)")
            records.append(o)
        return latents, records
