"""Host -> HBM hand-over of assembled batches (simpletuner_amd/training/collate.py::PinnedBatchStager / Prefetcher) on the MI355X: values arrive
bit-identical, floating fields are cast on the host side of the copy when asked, nested dicts and non-tensor entries pass through, pinned slabs
are reused (no per-step page-locking), and a staged batch drives prepare_batch unchanged."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _host_batch(i, B=2):
    g = torch.Generator().manual_seed(i)
    return {"latent_batch": torch.randn(B, 4, 16, 16, generator=g), "prompt_embeds": torch.randn(B, 9, 128, generator=g).to(BF16),
            "add_text_embeds": torch.randn(B, 64, generator=g).to(BF16), "batch_time_ids": torch.tensor([[[64.0, 48.0, 0.0, 0.0, 64.0, 48.0]]] * B),
            "extras": {"resolution": torch.tensor([[128, 128]] * B), "note": "kept"}, "prompts": ["a", "b"], "timesteps": torch.tensor([37, 811]),
            "conditioning_latents": None}


def test_stager_roundtrip_casts_and_reuses_pinned_slabs():
    from simpletuner_amd.training.collate import PinnedBatchStager
    dev = torch.device("cuda", 0)
    st = PinnedBatchStager(dev, slots=2)
    for i in range(5):
        host = _host_batch(i)
        out = PinnedBatchStager.wait(st.stage(host, dtype_map={"latent_batch": BF16, "batch_time_ids": BF16}))
        assert "_ready" not in out and out["prompts"] == ["a", "b"] and out["conditioning_latents"] is None and out["extras"]["note"] == "kept"
        assert out["latent_batch"].device == dev and out["latent_batch"].dtype == BF16
        assert torch.equal(out["latent_batch"].cpu(), host["latent_batch"].to(BF16))
        assert torch.equal(out["prompt_embeds"].cpu(), host["prompt_embeds"]) and torch.equal(out["add_text_embeds"].cpu(), host["add_text_embeds"])
        assert out["timesteps"].dtype == torch.int64 and torch.equal(out["timesteps"].cpu(), host["timesteps"])      # integer fields are never cast
        assert torch.equal(out["extras"]["resolution"].cpu(), host["extras"]["resolution"])
    slabs = [t for slot in st.slots for t in slot["pinned"].values()]
    assert len(slabs) == 2 * 6 and all(t.is_pinned() for t in slabs)       # 6 tensor fields x 2 slots, allocated once


def test_prefetched_staged_batches_drive_prepare_batch():
    from simpletuner_amd.sdxl.model import SDXL
    from simpletuner_amd.training.collate import PinnedBatchStager, Prefetcher
    from simpletuner_amd.training.trainer import St355Accelerator, default_config
    dev = torch.device("cuda", 0)
    pl = SDXL(default_config(model_family="sdxl", model_type="full"), St355Accelerator(dev))
    pl.setup_training_noise_schedule()
    feed = iter([_host_batch(i) for i in range(3)] + [False])
    pf = Prefetcher(lambda: next(feed), PinnedBatchStager(dev), depth=2, dtype_map={"latent_batch": BF16, "batch_time_ids": BF16})
    seen = 0
    while True:
        b = pf.next()
        if not b:
            break
        host = _host_batch(seen)
        pb = pl.prepare_batch(b, {"global_step": seen})
        assert pb["latents"].is_cuda and torch.equal(pb["latents"].cpu(), host["latent_batch"].to(BF16))
        assert pb["added_cond_kwargs"]["time_ids"].shape == (2, 1, 6) and pb["noisy_latents"].shape == (2, 4, 16, 16)
        assert torch.equal(pb["timesteps"].cpu(), torch.tensor([37, 811]))
        seen += 1
    assert seen == 3
    pf.close()
