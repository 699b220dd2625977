"""The reference's on-disk cache layout (simpletuner_amd/training/cache_io.py): file names equal the outputs of the reference's own
naming methods executed by tools/gen_golden.py::gen_cache_names (tests/golden/cache_io_vectors.pt), a gzip-wrapped payload produced by the
reference's compressor loads, and a cache directory in that layout feeds assemble_batch end to end."""
from pathlib import Path

import pytest
import torch

from simpletuner_amd.training import cache_io as IO
from simpletuner_amd.training.collate import assemble_batch

G = torch.load(Path(__file__).parent / "golden" / "cache_io_vectors.pt", weights_only=False)


def test_cache_file_names_match_reference_methods():
    assert len(G["vae"]) == 5 and len(G["text"]) == 5
    for fp, cache_dir, inst, hashed, want in G["vae"]:
        assert IO.vae_cache_filename(fp, cache_dir, inst, hashed) == tuple(want), fp
    for key, prompt, model_type, path_based, filename_key, want in G["text"]:
        got = IO.text_embed_cache_filename(key, "/c", model_type, prompt=prompt, path_based_keys=path_based, filename_key=filename_key)
        assert got == "/c/" + want + ".pt", (key, got, want)


def test_payloads_plain_and_gzip_roundtrip(tmp_path):
    obj = IO.loads_cache_payload(G["gz_bytes"])                               # bytes written by the reference's _compress_torch
    assert sorted(obj) == ["pooled_prompt_embeds", "prompt_embeds"] and torch.equal(obj["prompt_embeds"], G["gz_payload"]["prompt_embeds"])
    for compress in (False, True):
        p = tmp_path / f"x{int(compress)}" / "entry.pt"
        IO.save_cache_file(str(p), {"latents": torch.arange(6.0).reshape(1, 2, 3), "crop": (0, 8)}, compress=compress)
        raw = p.read_bytes()
        assert (raw[:2] == IO.GZIP_MAGIC) is compress and not list(p.parent.glob("*.tmp.*"))
        lat, meta = IO.latent_from_payload(IO.loads_cache_payload(raw))
        assert torch.equal(lat, torch.arange(6.0).reshape(1, 2, 3)) and meta == {"crop": (0, 8)}
    assert IO.latent_from_payload(torch.zeros(2))[1] is None
    with pytest.raises(ValueError, match="without 'latents'"):
        IO.latent_from_payload({"x": 1})
    with pytest.raises(FileNotFoundError):
        IO.CacheReader(str(tmp_path), str(tmp_path), "flux").read([{"image_path": "/nope/a.png", "instance_prompt_text": "a"}])


def test_cache_directory_feeds_assemble_batch(tmp_path):
    inst, vae_dir, txt_dir = "/data/imgs", str(tmp_path / "vae"), str(tmp_path / "text")
    examples = [dict(image_path=f"{inst}/set{i % 2}/img{i}.png", instance_prompt_text=f"caption {i}", crop_coordinates=(0, 0), intermediary_size=(1024, 1024),
                     aspect_ratio=1.0, data_backend_id="ds") for i in range(3)]
    for i, ex in enumerate(examples):
        path, _ = IO.vae_cache_filename(ex["image_path"], vae_dir, inst, hash_filenames=True)
        IO.save_cache_file(path, torch.full((16, 8, 8), float(i)).to(torch.bfloat16), compress=bool(i % 2))
        IO.save_cache_file(IO.text_embed_cache_filename(ex["instance_prompt_text"], txt_dir, "flux"),
                           {"prompt_embeds": torch.full((1, 5, 32), float(i)), "pooled_prompt_embeds": torch.full((8,), float(i))})
    rd = IO.CacheReader(vae_dir, txt_dir, "flux", instance_data_dir=inst, hash_filenames=True, workers=4)
    assert Path(rd.latent_path(examples[1]["image_path"])).parent.name == "set1"
    lats, recs = rd.read(examples)
    b = assemble_batch(examples, lats, recs, model_family="flux")
    assert b["latent_batch"].shape == (3, 16, 8, 8) and b["latent_batch"].dtype == torch.bfloat16 and b["prompt_embeds"].shape == (3, 5, 32)
    assert [b["latent_batch"][i, 0, 0, 0].item() for i in range(3)] == [0.0, 1.0, 2.0] and b["add_text_embeds"][2, 0].item() == 2.0
    assert b["batch_time_ids"] is None and b["prompts"] == ["caption 0", "caption 1", "caption 2"]
