"""LoRA key dialects (simpletuner_amd/training/lora_keys.py) against the reference's lora_format.py executed by
tools/gen_golden.py::gen_lora_keys (tests/golden/lora_keys_vectors.pt): detection, rank / alpha collection, LoraConfig kwargs, and every
converter — resulting key sets, tensor shapes and alpha values — plus ComfyUI export / import through the plugin's save / load."""
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch

from simpletuner_amd.training import lora_keys as LK

G = torch.load(Path(__file__).parent / "golden" / "lora_keys_vectors.pt", weights_only=False)


def _zoo():
    """the same dialect zoo tools/gen_golden.py::_lora_key_cases builds (only keys and shapes matter)"""
    def AB(r, i=16, o=16):
        return torch.zeros(r, i), torch.zeros(o, r)
    peft_tr, peft_unet, old, comfy = {}, {}, {}, {}
    for mod, r in (("transformer_blocks.0.attn.to_q", 4), ("transformer_blocks.0.attn.to_out.0", 4), ("single_transformer_blocks.3.attn.to_k", 8)):
        peft_tr[f"transformer.{mod}.lora_A.weight"], peft_tr[f"transformer.{mod}.lora_B.weight"] = AB(r)
    for mod in ("down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q", "mid_block.attentions.0.transformer_blocks.0.attn2.processor.to_v"):
        a, b = AB(4)
        peft_unet[f"unet.{mod}.lora_A.weight"], peft_unet[f"unet.{mod}.lora_B.weight"] = a, b
        old[f"unet.{mod}.lora.down.weight"], old[f"unet.{mod}.lora.up.weight"] = a, b
    peft_unet["text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_A.weight"], peft_unet["text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_B.weight"] = AB(4)
    peft_unet["text_encoder_2.text_model.encoder.layers.1.mlp.fc1.lora_A.weight"], peft_unet["text_encoder_2.text_model.encoder.layers.1.mlp.fc1.lora_B.weight"] = AB(4)
    peft_unet["unet.some.buffer"] = torch.zeros(2)
    for mod, r, al in (("double_blocks.0.img_attn.qkv", 4, 8.0), ("single_blocks.1.linear1", 8, 8.0)):
        comfy[f"diffusion_model.{mod}.lora_A.weight"], comfy[f"diffusion_model.{mod}.lora_B.weight"] = AB(r)
        comfy[f"diffusion_model.{mod}.alpha"] = torch.tensor(al)
    comfy["transformer.x.lora_A.weight"], comfy["bare.module.lora_B.weight"] = AB(4)
    return dict(peft_tr=peft_tr, peft_unet=peft_unet, old=old, comfy=comfy)


def _shape_of(d):
    return {k: (tuple(v.shape), (float(v) if v.ndim == 0 else None)) for k, v in d.items()}


def test_detection_collection_and_config_kwargs_match_reference():
    Z = _zoo()
    for name, d in dict(Z, empty={}).items():
        got = LK.detect_state_dict_format(d)
        assert (None if got is None else got.value) == G["detect"][name], name
    for v in (None, "", "ComfyUI ", "comfyui", "diffusers", "kohya", 3):
        assert LK.normalize_lora_format(v).value == G["normalize"][repr(v)]
    for name, d in Z.items():
        assert LK.collect_lora_ranks(d) == G["ranks"][name], name
        assert LK.collect_lora_alphas(d) == G["alphas"][name], name
        assert LK.synthesize_missing_lora_alphas_from_ranks(d) == G["synth"][name], name
        assert LK.peft_lora_config_kwargs_from_state_dict(d) == G["peft_kwargs"][name], name
    assert LK.collect_lora_ranks(Z["peft_tr"], prefix_to_strip="transformer.") == G["ranks_stripped"]
    assert LK.synthesize_missing_lora_alphas_from_ranks(Z["peft_tr"], existing_alphas={"x.alpha": 1.0}) == G["synth_existing"] == {}
    assert G["synth"]["peft_tr"] and G["peft_kwargs"]["peft_tr"]["rank_pattern"]            # the zoo does exercise the mixed-rank branches
    with pytest.raises(ValueError) as e:
        LK.collect_lora_ranks({"m.lora_A.weight": torch.zeros(4, 8), "m.lora_B.weight": torch.zeros(8, 2)})
    assert str(e.value) == G["conflict"]


def test_converters_match_reference_keys_shapes_and_alphas():
    Z = _zoo()
    meta = {"lora_alpha": 16, "alpha_pattern": {"single_transformer_blocks.3.attn.to_k": 2.0}}
    assert _shape_of(LK.convert_diffusers_to_comfyui(Z["peft_tr"])) == G["to_comfy"]
    assert _shape_of(LK.convert_diffusers_to_comfyui(Z["peft_tr"], adapter_metadata=meta, preserve_component_prefixes={"transformer"})) == G["to_comfy_keep_meta"]
    assert _shape_of(LK.convert_diffusers_to_comfyui(Z["old"], adapter_metadata={"lora_alpha": torch.tensor(4.0)})) == G["to_comfy_old"]
    assert _shape_of(LK.convert_diffusers_to_comfyui_sd_lora(Z["peft_unet"], adapter_metadata={"lora_alpha": 8},
                                                             component_adapter_metadata={"text_encoder": {"lora_alpha": 2}}, sdxl=True)) == G["to_kohya_sdxl"]
    assert _shape_of(LK.convert_diffusers_to_comfyui_sd_lora(Z["old"], sdxl=False)) == G["to_kohya_sd15"]
    sd, al = LK.convert_comfyui_to_diffusers(Z["comfy"], target_prefix="transformer")
    assert (_shape_of(sd), al) == G["from_comfy"]
    sd, al = LK.convert_comfyui_to_diffusers(Z["comfy"])
    assert (_shape_of(sd), al) == G["from_comfy_noprefix"]
    assert any(k.startswith("lora_unet_mid_block_attentions_0_transformer_blocks_0_attn2_to_v.") for k in G["to_kohya_sdxl"])      # `.processor.` dropped
    assert any(k.startswith("lora_te1_") for k in G["to_kohya_sdxl"]) and any(k.startswith("lora_te2_") for k in G["to_kohya_sdxl"])


def test_comfyui_export_and_import_through_the_plugin(tmp_path):
    """config.lora_format = "comfyui": the file holds `diffusion_model.` keys + `.alpha` tensors (or keeps `transformer.` for families that preserve
    it), and load_lora_weights reads either dialect back into the adapters"""
    from safetensors.torch import load_file

    from simpletuner_amd.foundation import ModelFoundation

    class Comp(torch.nn.Module):
        def __init__(self):
            super().__init__()
            mk = lambda i, o: torch.nn.ModuleDict({"default": torch.nn.Linear(i, o, bias=False)})
            self.blk = torch.nn.ModuleDict({"to_q": torch.nn.ModuleDict({"lora_A": mk(8, 2), "lora_B": mk(2, 8)})})

    for subfolder, keep, want_prefix in (("unet", set(), "diffusion_model."), ("transformer", {"transformer"}, "transformer.")):
        class Plug(ModelFoundation):
            MODEL_SUBFOLDER = subfolder
            COMFYUI_LORA_PRESERVE_COMPONENT_PREFIXES = keep
        m = Plug(SimpleNamespace(lora_format="comfyui", lora_alpha=4.0, lora_rank=2), SimpleNamespace(device=torch.device("cpu")))
        m.model = Comp()
        d = tmp_path / subfolder
        flat = load_file(m.save_lora_weights(str(d)))
        assert sorted(flat) == [f"{want_prefix}blk.to_q.alpha", f"{want_prefix}blk.to_q.lora_A.weight", f"{want_prefix}blk.to_q.lora_B.weight"]
        assert flat[f"{want_prefix}blk.to_q.alpha"].item() == 4.0
        want = m.model.blk["to_q"]["lora_A"]["default"].weight.detach().clone()
        with torch.no_grad():
            m.model.blk["to_q"]["lora_A"]["default"].weight.zero_()
        m.load_lora_weights(input_dir=str(d))
        assert torch.equal(m.model.blk["to_q"]["lora_A"]["default"].weight, want)


def test_sd_family_kohya_export_and_import(tmp_path):
    """SDXL / SD1.x with lora_format = "comfyui": kohya names on disk (sdxl/model.py:61-75), mapped back onto the adapters on load; a file whose
    alpha disagrees with the configured adapter scale is refused"""
    from safetensors.torch import load_file, save_file

    from simpletuner_amd.sdxl.model import SDXL

    class Comp(torch.nn.Module):
        def __init__(self):
            super().__init__()
            mk = lambda i, o: torch.nn.ModuleDict({"default": torch.nn.Linear(i, o, bias=False)})
            leaf = lambda: torch.nn.ModuleDict({"lora_A": mk(8, 2), "lora_B": mk(2, 8)})
            self.down_blocks = torch.nn.ModuleDict({"1": torch.nn.ModuleDict({"attn1": torch.nn.ModuleDict({"to_q": leaf(), "to_out": torch.nn.ModuleDict({"0": leaf()})})})})

    m = SDXL(SimpleNamespace(lora_format="comfyui", lora_alpha=None, lora_rank=2), SimpleNamespace(device=torch.device("cpu")))
    m.model = Comp()
    path = m.save_lora_weights(str(tmp_path))
    flat = load_file(path)
    assert sorted(flat) == ["lora_unet_down_blocks_1_attn1_to_out_0.alpha", "lora_unet_down_blocks_1_attn1_to_out_0.lora_down.weight",
                            "lora_unet_down_blocks_1_attn1_to_out_0.lora_up.weight", "lora_unet_down_blocks_1_attn1_to_q.alpha",
                            "lora_unet_down_blocks_1_attn1_to_q.lora_down.weight", "lora_unet_down_blocks_1_attn1_to_q.lora_up.weight"]
    assert flat["lora_unet_down_blocks_1_attn1_to_q.alpha"].item() == 2.0                      # alpha defaults to the rank
    w = m.model.down_blocks["1"]["attn1"]["to_out"]["0"]["lora_B"]["default"].weight
    want = w.detach().clone()
    with torch.no_grad():
        w.zero_()
    m.load_lora_weights(input_dir=str(tmp_path))
    assert torch.equal(w, want)
    flat["lora_unet_down_blocks_1_attn1_to_q.alpha"] = torch.tensor(16.0)
    save_file(flat, path)
    with pytest.raises(ValueError, match="differs from the configured lora_alpha"):
        m.load_lora_weights(input_dir=str(tmp_path))
