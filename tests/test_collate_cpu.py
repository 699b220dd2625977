"""Batch assembly (simpletuner_amd/training/collate.py) against outputs of the reference's own functions
(tests/golden/collate_vectors.pt, tools/gen_golden.py::gen_collate executes collate.py:59-98 and :501-523 as written) and the reference's
known-answer tests for tensor collation (tests/test_collate_dimensions.py:392-447)."""
from pathlib import Path

import pytest
import torch

from simpletuner_amd.training import collate as C

G = torch.load(Path(__file__).parent / "golden" / "collate_vectors.pt", weights_only=False)


def test_time_ids_match_reference_code_outputs():
    assert len(G["time_ids"]) == 6
    for inter, tgt, crop, dt, want in G["time_ids"]:
        got = C.compute_time_ids(inter, tgt, dt, crop_coordinates=list(crop))
        assert got.dtype == want.dtype and torch.equal(got, want)
    examples, lat_shape, want = G["sdxl"]
    got = C.gather_conditional_sdxl_size_features(examples, torch.zeros(lat_shape), torch.bfloat16)
    assert got.shape == (3, 1, 6) and torch.equal(got, want) and torch.count_nonzero(got[1]) == 0      # dropped conditioning -> zeros
    with pytest.raises(ValueError, match="Crop coordinates"):
        C.compute_time_ids((1024, 1024), (4, 128, 128), torch.float32)
    with pytest.raises(ValueError, match="must match"):
        C.gather_conditional_sdxl_size_features(examples[:2], torch.zeros(lat_shape), torch.bfloat16)
    assert C.compute_time_ids((1024, 768), (4, 96, 128), torch.float32, crop_coordinates=(0, 0), refiner_aesthetic_score=6.0).tolist() == [[768, 1024, 0, 0, 6.0]]


def test_pixart_size_features():
    f = C.gather_conditional_pixart_size_features([{}, {}], torch.zeros(2, 4, 256, 128), torch.bfloat16)
    assert f["resolution"].tolist() == [[2048, 1024]] * 2 and f["aspect_ratio"].tolist() == [[2.0]] * 2 and f["resolution"].dtype == torch.bfloat16


def test_collate_tensors_known_answers():
    """tests/test_collate_dimensions.py:392-447"""
    assert C.collate_tensors([torch.randn(256, 2048) for _ in range(4)]).shape == (4, 256, 2048)
    assert C.collate_tensors([torch.randn(1, 256, 2048) for _ in range(4)]).shape == (4, 256, 2048)
    assert C.collate_tensors([torch.randn(256) for _ in range(4)]).shape == (4, 256)
    mixed = [torch.randn(1, 256, 2048), torch.randn(256, 2048), torch.randn(1, 256, 2048), torch.randn(256, 2048)]
    assert C.collate_tensors(mixed).shape == (4, 256, 2048)
    assert C.collate_tensors([]) is None and C.collate_tensors([None]) is None
    assert C.collate_tensors([torch.randn(256, 2048)]).shape == (1, 256, 2048)
    assert C.collate_tensors([torch.randn(1, 256, 2048)]).shape == (1, 256, 2048)
    assert C.collate_tensors([torch.randn(2, 256, 2048), torch.randn(3, 256, 2048)]).shape == (5, 256, 2048)
    with pytest.raises(ValueError, match="Unexpected tensor dimension"):
        C.collate_tensors([torch.randn(1, 4, 4), torch.randn(1, 1, 4, 4)])


def test_collate_prompt_embeds_key_mapping_and_model_hook():
    recs = [{"prompt_embeds": torch.randn(1, 7, 16), "pooled_prompt_embeds": torch.randn(8), "attention_mask": torch.ones(7)} for _ in range(3)]      # 2-D entries are STACKED (a [1, D] pooled row becomes [B, 1, D], as in the reference)
    out = C.collate_prompt_embeds(recs)
    assert out["prompt_embeds"].shape == (3, 7, 16) and out["pooled_prompt_embeds"].shape == (3, 8) and out["attention_masks"].shape == (3, 7)
    assert C.collate_prompt_embeds([{"prompt_attention_mask": torch.ones(7), "prompt_embeds": torch.randn(7, 16)}])["attention_masks"].shape == (1, 7)

    class Own:
        def collate_prompt_embeds(self, recs):
            return {"prompt_embeds": torch.zeros(len(recs), 1, 1)}
    assert C.collate_prompt_embeds(recs, Own())["prompt_embeds"].shape == (3, 1, 1)
    with pytest.raises(Exception, match="Could not compute text encoder output"):
        C.collate_prompt_embeds([{"unknown": 1}])


def test_check_latent_shapes_rules():
    lats = [torch.randn(4, 8, 8) for _ in range(3)]
    fps = ["a.png", "b.png", "c.png"]
    ex = [{"aspect_ratio": 1.0}] * 3
    assert C.check_latent_shapes(lats, fps, "ds", ex).shape == (3, 4, 8, 8)
    slab = torch.empty(3, 4, 8, 8)
    assert C.check_latent_shapes(lats, fps, "ds", ex, out=slab) is slab and torch.equal(slab[2], lats[2])
    with pytest.raises(ValueError, match="Aspect ratio mismatch"):
        C.check_latent_shapes(lats, fps, "ds", [{"aspect_ratio": 1.0}, {"aspect_ratio": 1.5}, {"aspect_ratio": 1.0}])
    with pytest.raises(ValueError, match="b.png latent is None"):
        C.check_latent_shapes([lats[0], None, lats[2]], fps, "ds", ex)
    bad, seen = lats[1].clone(), []
    bad[0, 0, 0] = float("nan")
    with pytest.raises(ValueError, match="contains NaN or Inf"):
        C.check_latent_shapes([lats[0], bad, lats[2]], fps, "ds", ex, on_corrupt=seen.append)
    assert seen == ["b.png"]
    with pytest.raises(ValueError, match="latent shape mismatch"):
        C.check_latent_shapes([lats[0], torch.randn(4, 8, 16), lats[2]], fps, "ds", ex)
    ragged = C.check_latent_shapes([lats[0], torch.randn(4, 8, 16)], fps[:2], "ds", ex[:2], is_conditioning=True)    # ControlNet inputs may differ
    assert isinstance(ragged, list) and len(ragged) == 2


def test_assemble_batch_keys_dropout_and_sdxl_time_ids():
    B = 3
    examples = [dict(image_path=f"{i}.png", instance_prompt_text=f"cap {i}", crop_coordinates=(0, 8 * i), intermediary_size=(1024 + 8 * i, 1024),
                     aspect_ratio=1.0, data_backend_id="ds1") for i in range(B)]
    lats = [torch.randn(4, 128, 128).to(torch.bfloat16) for _ in range(B)]
    recs = [dict(prompt_embeds=torch.full((1, 77, 32), float(i + 1)), pooled_prompt_embeds=torch.full((16,), float(i + 1))) for i in range(B)]
    empty = dict(prompt_embeds=torch.zeros(1, 77, 32), pooled_prompt_embeds=torch.zeros(16))
    draws = iter([0.9, 0.05, 0.5])                                          # only example 1 falls under p = 0.1
    b = C.assemble_batch(examples, lats, recs, model_family="sdxl", caption_dropout_probability=0.1, empty_prompt_record=empty, draw=lambda: next(draws))
    for key in ("latent_batch", "latent_metadata", "filepaths", "data_backend_id", "prompts", "text_encoder_output", "prompt_embeds", "add_text_embeds",
                "batch_time_ids", "encoder_attention_mask", "conditioning_latents", "conditioning_pixel_values", "conditioning_type", "loss_mask_type",
                "is_regularisation_data", "is_i2v_data"):                  # the image-model subset of collate.py:1316-1350
        assert key in b, key
    assert b["latent_batch"].shape == (B, 4, 128, 128) and b["prompt_embeds"].shape == (B, 77, 32) and b["add_text_embeds"].shape == (B, 16)
    assert b["prompts"] == ["cap 0", "", "cap 2"] and [e["drop_conditioning"] for e in examples] == [False, True, False]
    assert b["prompt_embeds"][1].abs().sum() == 0 and b["prompt_embeds"][2, 0, 0] == 3.0
    assert b["batch_time_ids"].shape == (B, 1, 6) and b["batch_time_ids"][1].abs().sum() == 0
    assert b["batch_time_ids"][2, 0].tolist() == [1024, 1040, 0, 16, 1024, 1024]
    assert b["encoder_attention_mask"] is None and b["data_backend_id"] == "ds1" and b["filepaths"] == ["0.png", "1.png", "2.png"]
    with pytest.raises(ValueError, match="empty prompt"):
        C.assemble_batch(examples, lats, recs, caption_dropout_probability=1.0, draw=lambda: 0.0)
    with pytest.raises(ValueError, match="must match"):
        C.assemble_batch(examples, lats[:2], recs)
    # ControlNet: conditioning latents ride along, count must match (collate.py:606-611)
    pb = C.assemble_batch(examples, lats, [dict(prompt_embeds=torch.zeros(300, 8), attention_masks=torch.ones(300)) for _ in range(B)], model_family="pixart_sigma",
                          conditioning_latents=[torch.zeros(4, 128, 128)] * B)
    assert pb["conditioning_latents"].shape == (B, 4, 128, 128) and pb["encoder_attention_mask"].shape == (B, 300)
    assert pb["batch_time_ids"]["resolution"].tolist() == [[1024, 1024]] * B
    with pytest.raises(ValueError, match="must match for ControlNet"):
        C.assemble_batch(examples, lats, recs, conditioning_latents=[torch.zeros(4, 128, 128)])


def test_stager_has_no_cpu_path():
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by tests/test_collate_gpu.py")
    with pytest.raises(RuntimeError, match="no CPU path"):
        C.PinnedBatchStager("cuda:0")


def test_prefetcher_delivers_in_order_and_surfaces_errors():
    items = iter([{"i": 0}, {"i": 1}, {"i": 2}, False])
    pf = C.Prefetcher(lambda: next(items), stager=None, depth=2)
    assert [pf.next() for _ in range(4)] == [{"i": 0}, {"i": 1}, {"i": 2}, False]      # the falsy epoch-end sentinel is delivered, then the thread stops
    pf.close()

    def boom():
        raise OSError("cache read failed")
    pf2 = C.Prefetcher(boom, stager=None)
    with pytest.raises(OSError, match="cache read failed"):
        pf2.next()
