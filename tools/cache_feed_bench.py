#!/usr/bin/env python3
"""Throughput of the cache feed: synthetic Flux-shaped cache files in the reference's layout (latents [16,128,128] bf16 = 512 KiB,
text record {prompt_embeds [1,512,4096], pooled_prompt_embeds [1,768], attention_masks [1,512]} = 4 MiB per caption) on local disk, read back through
  (a) DirectCacheFeeder (mmap -> pinned slab -> async DMA, one host copy),
  (b) CacheReader.read -> assemble_batch -> PinnedBatchStager (the collate mirror: unpickle + stack + pinned copy),
and compared with what one MI355X / an 8-GPU node consumes at the measured step rate.  Prints one JSON line.
    python tools/cache_feed_bench.py [--samples 256] [--batch 8] [--workers 16]
"""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    from simpletuner_amd.training.cache_feed import DirectCacheFeeder
    from simpletuner_amd.training.cache_io import CacheReader, save_cache_file
    from simpletuner_amd.training.collate import PinnedBatchStager, assemble_batch
    dev = torch.device("cuda", 0)
    root = a.dir or tempfile.mkdtemp(prefix="st355_cache_")
    vae_dir, txt_dir = os.path.join(root, "vae"), os.path.join(root, "text")
    rd = CacheReader(vae_dir, txt_dir, "flux", workers=a.workers)
    g = torch.Generator().manual_seed(0)
    examples = []
    for i in range(a.samples):
        ex = {"image_path": f"/data/img_{i:05d}.png", "instance_prompt_text": f"a photo number {i}", "aspect_ratio": 1.0, "data_backend_id": "bench"}
        save_cache_file(rd.latent_path(ex["image_path"]), torch.randn(16, 128, 128, generator=g).to(torch.bfloat16))
        save_cache_file(rd.text_path(ex["instance_prompt_text"]), {"prompt_embeds": torch.randn(1, 512, 4096, generator=g).to(torch.bfloat16),
                                                                   "pooled_prompt_embeds": torch.randn(768, generator=g).to(torch.bfloat16),
                                                                   "attention_masks": torch.ones(1, 512, dtype=torch.int64)})
        examples.append(ex)
    per_image = 16 * 128 * 128 * 2 + 512 * 4096 * 2 + 768 * 2 + 512 * 8
    batches = [examples[i:i + a.batch] for i in range(0, a.samples, a.batch)]

    def run_direct():
        feed = DirectCacheFeeder(rd, dev, workers=a.workers)
        for b in batches:
            feed.submit(b)
        feed.close()
        t0 = time.perf_counter()
        n, chk = 0, 0.0
        while True:
            b = feed.next()
            if b is None:
                break
            n += b["latent_batch"].shape[0]
            chk += float(b["latent_batch"][0, 0, 0, 0])            # forces the batch to be consumed on the device
        torch.cuda.synchronize()
        return n, time.perf_counter() - t0

    def run_collate():
        st = PinnedBatchStager(dev, slots=3)
        t0 = time.perf_counter()
        n = 0
        for b in batches:
            lat, recs = rd.read(b)
            batch = assemble_batch([dict(e) for e in b], lat, recs, model_family="flux")
            out = PinnedBatchStager.wait(st.stage(batch))
            n += out["latent_batch"].shape[0]
        torch.cuda.synchronize()
        return n, time.perf_counter() - t0

    run_direct()                                                   # warm the page cache and the pinned allocator for both paths
    n1, t1 = run_direct()
    n2, t2 = run_collate()
    step_rate = 5.4                                                # images/s of one MI355X on the Flux LoRA step (profiles/archive/r02_bench_line.json)
    print(json.dumps({"what": "cache feed throughput, Flux-shaped latent + text-embed cache files on local disk (page cache warm)", "samples": a.samples,
                      "batch": a.batch, "workers": a.workers, "bytes_per_image": per_image,
                      "direct_feeder": {"images_per_s": round(n1 / t1, 1), "GB_per_s": round(n1 * per_image / t1 / 1e9, 2)},
                      "collate_mirror": {"images_per_s": round(n2 / t2, 1), "GB_per_s": round(n2 * per_image / t2 / 1e9, 2)},
                      "consumption": {"one_gpu_images_per_s": step_rate, "eight_gpu_node_images_per_s": 8 * step_rate,
                                      "headroom_direct_vs_node": round(n1 / t1 / (8 * step_rate), 1)}, "host_cores": os.cpu_count()}))


if __name__ == "__main__":
    main()
