"""DirectCacheFeeder: the one-copy cache feed delivers exactly what the collate mirror (CacheReader -> assemble_batch -> PinnedBatchStager)
delivers for the same examples, from plain and gzip-wrapped cache files, and refuses corrupt / mis-shaped entries like the reference's collate does."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _write(rd, n, compress=False, bad=None):
    from simpletuner_amd.training.cache_io import save_cache_file
    g = torch.Generator().manual_seed(3)
    exs = []
    for i in range(n):
        ex = {"image_path": f"/data/i{i}.png", "instance_prompt_text": f"caption {i}", "aspect_ratio": 1.0, "data_backend_id": "t"}
        lat = torch.randn(16, 8, 12, generator=g).to(torch.bfloat16)
        if bad == ("nan", i):
            lat[0, 0, 0] = float("nan")
        if bad == ("shape", i):
            lat = lat[:, :4]
        save_cache_file(rd.latent_path(ex["image_path"]), lat, compress=compress)
        save_cache_file(rd.text_path(ex["instance_prompt_text"]), {"prompt_embeds": torch.randn(1, 32, 64, generator=g).to(torch.bfloat16),
                                                                   "pooled_prompt_embeds": torch.randn(24, generator=g).to(torch.bfloat16),
                                                                   "attention_masks": (torch.rand(1, 32, generator=g) > 0.3).long()}, compress=compress)
        exs.append(ex)
    return exs


@pytest.mark.parametrize("compress", [False, True])
def test_direct_feed_equals_the_collate_mirror(tmp_path, compress):
    from simpletuner_amd.training.cache_feed import DirectCacheFeeder
    from simpletuner_amd.training.cache_io import CacheReader
    from simpletuner_amd.training.collate import PinnedBatchStager, assemble_batch
    dev = torch.device("cuda", 0)
    rd = CacheReader(str(tmp_path / "vae"), str(tmp_path / "text"), "flux")
    exs = _write(rd, 12, compress=compress)
    feed = DirectCacheFeeder(rd, dev, workers=4, slots=2)
    batches = [exs[0:4], exs[4:8], exs[8:12], exs[0:4]]              # 4 batches over 2 slots: slab reuse is exercised
    for b in batches:
        feed.submit(b)
    feed.close()
    st = PinnedBatchStager(dev)
    for b in batches:
        got = feed.next()
        lat, recs = rd.read(b)
        want = PinnedBatchStager.wait(st.stage(assemble_batch([dict(e) for e in b], lat, recs, model_family="flux")))
        torch.cuda.synchronize()
        assert torch.equal(got["latent_batch"], want["latent_batch"]) and got["latent_batch"].shape == (4, 16, 8, 12)
        assert torch.equal(got["prompt_embeds"], want["prompt_embeds"]) and torch.equal(got["add_text_embeds"], want["add_text_embeds"])
        assert torch.equal(got["encoder_attention_mask"], want["encoder_attention_mask"].to(got["encoder_attention_mask"].dtype))
        assert got["filepaths"] == want["filepaths"] and got["prompts"] == want["prompts"]
    assert feed.next() is None


@pytest.mark.parametrize("bad,msg", [(("nan", 2), "NaN or Inf"), (("shape", 1), "shape mismatch")])
def test_direct_feed_refuses_corrupt_entries(tmp_path, bad, msg):
    from simpletuner_amd.training.cache_feed import DirectCacheFeeder
    from simpletuner_amd.training.cache_io import CacheReader
    rd = CacheReader(str(tmp_path / "vae"), str(tmp_path / "text"), "flux")
    exs = _write(rd, 4, bad=bad)
    feed = DirectCacheFeeder(rd, torch.device("cuda", 0), workers=2)
    feed.submit(exs)
    with pytest.raises(ValueError, match=msg):
        feed.next()
