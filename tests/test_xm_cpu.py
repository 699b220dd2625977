"""XM (explorative modelling with noise candidates; off by default): the candidate-major batch algebra and the configuration rules, ported from
the reference's tests/test_sd3_model.py:29-110 and tests/test_flux_model.py:93-131 onto simpletuner_amd/xm.py.  The loss itself runs on the HIP
loss kernel (tests/test_xm_gpu.py); here the noising kernels are replaced by their torch formulas so the host logic is checked without a GPU."""
from types import SimpleNamespace
from unittest.mock import patch

import pytest
import torch

from simpletuner_amd import ops
from simpletuner_amd.foundation import ModelFoundation, PredictionTypes
from simpletuner_amd.xm import (ExplorativeModelingConfig, reshape_candidate_batch, route_usage_histogram, select_min_candidate_loss,
                                select_winning_candidates, winner_weights)


def _plugin(**cfg):
    base = dict(xm_enabled=True, xm_candidate_count=2, xm_training_target="noise", xm_selection_scope="sample", xm_block_size=0)
    base.update(cfg)
    m = ModelFoundation(SimpleNamespace(**base), SimpleNamespace(device=torch.device("cpu")))
    m.NAME = "Plug"
    return m


def _torch_flow_mix(lat, sig, noise=None, seed=0, offset=0):
    s = sig.reshape(-1, *([1] * (lat.dim() - 1))).to(lat)
    return (1.0 - s) * lat + s * noise, noise - lat, noise


def test_config_rules():
    assert ExplorativeModelingConfig.from_config(SimpleNamespace()).enabled is False
    assert ExplorativeModelingConfig.from_config({"xm_enabled": True, "xm_candidate_count": 3}).candidate_count == 3
    for bad in (dict(xm_enabled=True, xm_candidate_count=1), dict(xm_training_target="x"), dict(xm_selection_scope="x"), dict(xm_block_size=-1),
                dict(xm_selection_scope="block", xm_block_size=1)):
        with pytest.raises(ValueError):
            ExplorativeModelingConfig.from_config(bad)
    with pytest.raises(ValueError, match="xm_training_target='noise'"):                       # test_sd3_model.py:29-41
        _plugin(xm_training_target="route")._xm_noise_candidates_enabled({})
    with pytest.raises(ValueError, match="xm_selection_scope='sample'"):                      # :43-54
        _plugin(xm_selection_scope="block", xm_block_size=2)._xm_noise_candidates_enabled({})
    with pytest.raises(ValueError, match="input_perturbation"):
        _plugin(input_perturbation=0.1)._xm_noise_candidates_enabled({})
    m = _plugin()
    assert m._xm_noise_candidates_enabled({}) is True
    assert m._xm_noise_candidates_enabled({"xm_candidate_count": 2}) is False                 # already expanded
    assert m._xm_noise_candidates_enabled({"xm_winner_indices": torch.tensor([0])}) is False  # already cut back
    assert _plugin(xm_enabled=False)._xm_noise_candidates_enabled({}) is False


def test_candidate_algebra():
    losses = torch.tensor([5.0, 1.0, 2.0, 4.0, 0.5, 9.0])                                     # K=2, B=3, candidate-major
    cand = reshape_candidate_batch(losses, 2)
    sel, who = select_min_candidate_loss(cand)
    assert who.tolist() == [1, 1, 0] and sel.item() == pytest.approx((4.0 + 0.5 + 2.0) / 3)
    vals = torch.arange(6 * 2, dtype=torch.float32).reshape(6, 2)
    assert torch.equal(select_winning_candidates(vals, who, 2), vals[[3, 4, 2]])
    assert route_usage_histogram(who, 2).tolist() == [1.0, 2.0]
    w = winner_weights(who, 2)
    assert w.tolist() == [0.0, 0.0, 2.0, 2.0, 2.0, 0.0]
    assert (w * losses).mean().item() == pytest.approx(sel.item())                           # mean over K*B weighted rows == mean over winners
    base = torch.tensor([1.0, 2.0, 3.0, 1.0, 2.0, 3.0])
    assert winner_weights(who, 2, base).tolist() == [0.0, 0.0, 6.0, 2.0, 4.0, 0.0]
    with pytest.raises(ValueError):
        reshape_candidate_batch(torch.zeros(5), 2)


def test_prepare_xm_noise_candidates_expands_candidate_major():
    """tests/test_sd3_model.py:56-102"""
    torch.manual_seed(1)
    m = _plugin()
    lat = torch.arange(64, dtype=torch.float32).reshape(2, 2, 4, 4)
    pb = {"latents": lat.clone(), "noise": torch.zeros_like(lat), "input_noise": torch.zeros_like(lat), "noisy_latents": lat.clone(),
          "sigmas": torch.tensor([0.25, 0.75]).view(2, 1, 1, 1), "timesteps": torch.tensor([250.0, 750.0]),
          "encoder_hidden_states": torch.arange(12, dtype=torch.float32).reshape(2, 3, 2), "add_text_embeds": torch.arange(8, dtype=torch.float32).reshape(2, 4),
          "added_cond_kwargs": {"text_embeds": torch.arange(8, dtype=torch.float32).reshape(2, 4)},
          "conditioning_latents": torch.ones(2, 2, 4, 4), "flowmap_r_timesteps": torch.tensor([125.0, 375.0]), "metadata": [{"id": 0}, {"id": 1}],
          "scalar": torch.tensor(3.0)}
    with patch.object(ops, "flow_noise_mix", _torch_flow_mix):
        m._prepare_xm_noise_candidates(pb)
    assert pb["latents"].shape[0] == 4 and torch.equal(pb["latents"][:2], lat) and torch.equal(pb["latents"][2:], lat)
    for k in ("encoder_hidden_states", "add_text_embeds", "conditioning_latents"):
        assert torch.equal(pb[k][:2], pb[k][2:])
    assert torch.equal(pb["added_cond_kwargs"]["text_embeds"][:2], pb["added_cond_kwargs"]["text_embeds"][2:])
    assert torch.equal(pb["flowmap_r_timesteps"], torch.tensor([125.0, 375.0, 125.0, 375.0]))
    assert pb["metadata"] == [{"id": 0}, {"id": 1}, {"id": 0}, {"id": 1}]
    assert pb["scalar"].ndim == 0
    assert not torch.equal(pb["noise"][:2], pb["noise"][2:]) and pb["input_noise"] is pb["noise"]
    assert torch.equal(pb["flow_target"], pb["noise"] - pb["latents"])
    assert torch.equal(pb["noisy_latents"], (1.0 - pb["sigmas"]) * pb["latents"] + pb["sigmas"] * pb["noise"])
    assert pb["xm_candidate_count"] == 2 and pb["xm_original_batch_size"] == 2
    # guards (xm_mixin.py:97-108)
    with pytest.raises(ValueError, match="explicit prepared target"):
        m._prepare_xm_noise_candidates({"latents": lat, "timesteps": torch.zeros(2), "noisy_latents": lat, "target": lat})
    with pytest.raises(ValueError, match="noisy_latents"):
        m._prepare_xm_noise_candidates({"latents": lat, "timesteps": torch.zeros(2)})


def test_winner_selection_shrinks_both_dictionaries():
    """tests/test_sd3_model.py:104-147 / tests/test_flux_model.py:132-165 (everything except the loss value, which is the kernel's)"""
    m = _plugin()
    who = torch.tensor([1, 0])
    hidden = torch.arange(4 * 3 * 2, dtype=torch.float32).reshape(4, 3, 2)
    pb = {"latents": torch.zeros(4, 1, 1, 4), "noise": torch.arange(4.0).view(4, 1, 1, 1), "timesteps": torch.ones(4),
          "metadata": [{"id": 0}, {"id": 1}, {"id": 0}, {"id": 1}], "xm_candidate_count": 2, "xm_original_batch_size": 2,
          "conditioning_packed_latents": torch.arange(4 * 2 * 3, dtype=torch.float32).view(4, 2, 3)}
    out = {"model_prediction": torch.arange(16.0).view(4, 1, 1, 4), "hidden_states_buffer": {"layer_2": hidden.clone()}, "crepa_hidden_states": None,
           "xm_candidate_count": 2}
    m._select_xm_winners_in_place(pb, out, who, 2)
    assert pb["latents"].shape[0] == 2 and pb["metadata"] == [{"id": 0}, {"id": 1}] and "xm_candidate_count" not in pb
    assert torch.equal(pb["noise"].flatten(), torch.tensor([2.0, 1.0])) and tuple(pb["conditioning_packed_latents"].shape) == (2, 2, 3)
    assert torch.equal(pb["xm_winner_indices"], who) and torch.equal(out["xm_winner_indices"], who) and "xm_candidate_count" not in out
    assert out["model_prediction"].shape[0] == 2 and torch.equal(out["hidden_states_buffer"]["layer_2"], hidden[[2, 1]])
    logs = m._xm_candidate_logs(torch.tensor(0.25), torch.tensor([[1.0, 0.5], [0.0, 2.0]]), who, 2)
    assert logs == {"xm_loss": 0.25, "xm_candidate_loss_mean": 0.875, "xm_candidate_0_wins": 1.0, "xm_candidate_1_wins": 1.0}


def test_model_predict_expands_then_marks_output():
    """flux/model.py:630-636: expansion happens inside model_predict, once, and the output carries the candidate count"""
    m = _plugin()
    m.PREDICTION_TYPE = PredictionTypes.FLOW_MATCHING
    seen = {}

    def single(pb):
        seen["B"] = pb["noisy_latents"].shape[0]
        return {"model_prediction": pb["noisy_latents"].clone()}

    m._model_predict_single = single
    lat = torch.randn(3, 4, 2, 2)
    pb = {"latents": lat, "noisy_latents": lat.clone(), "sigmas": torch.full((3, 1, 1, 1), 0.5), "timesteps": torch.full((3,), 500.0)}
    with patch.object(ops, "flow_noise_mix", _torch_flow_mix):
        out = m.model_predict(pb)
        assert seen["B"] == 6 and out["xm_candidate_count"] == 2 and pb["latents"].shape[0] == 6
        out2 = m.model_predict(pb)                                   # already expanded: passes through
    assert seen["B"] == 6 and "xm_candidate_count" not in out2
    off = _plugin(xm_enabled=False)
    off._model_predict_single = single
    off.model_predict({"noisy_latents": lat})
    assert seen["B"] == 3


def test_flux_guidance_scales_modes_and_xm_replication():
    """flux/model.py:682-705; the XM case is the reference's tests/test_flux_model.py:93-128 (guidance [1.25, 1.75, 1.25, 1.75])"""
    from simpletuner_amd.flux.model import Flux
    m = Flux.__new__(Flux)
    m.config = SimpleNamespace(flux_guidance_mode="constant", flux_guidance_value=3.5)
    assert m._flux_guidance_scales({}, 3) == [3.5, 3.5, 3.5]
    m.config = SimpleNamespace(flux_guidance_mode="random-range", flux_guidance_min=1.0, flux_guidance_max=2.0)
    with patch("random.uniform", side_effect=[1.25, 1.75]):
        assert m._flux_guidance_scales({"xm_candidate_count": 2, "xm_original_batch_size": 2}, 4) == [1.25, 1.75, 1.25, 1.75]
    vals = m._flux_guidance_scales({}, 5)
    assert len(vals) == 5 and all(1.0 <= v <= 2.0 for v in vals) and len(set(vals)) > 1
    m.config = SimpleNamespace(flux_guidance_mode="bogus")
    with pytest.raises(ValueError, match="Unsupported Flux guidance mode: 'bogus'"):
        m._flux_guidance_scales({}, 1)


def test_flux_lora_target_sets_are_exact_or_refused():
    """flux/model.py:1235-1380: 'all', 'context', the '+ffs' sets, 'tiny' / 'nano' and the fall-through default are built; every other named set is refused
    instead of being narrowed silently"""
    from simpletuner_amd.flux.model import Flux
    m = Flux.__new__(Flux)
    m.config = SimpleNamespace(flux_lora_target="all")
    assert m._lora_target_set() == "all" and "to_add_out" in m.get_lora_target_layers() and "add_q_proj" in m.get_lora_target_layers()
    for v in ("default", None, "mmdit", "something-else"):                       # unknown names fall through to DEFAULT_LORA_TARGET, as in the reference
        m.config = SimpleNamespace(flux_lora_target=v)
        assert m._lora_target_set() == "default" and m.get_lora_target_layers() == ["to_k", "to_q", "to_v", "to_out.0"]
    m.config = SimpleNamespace(flux_lora_target="context")                       # flux/model.py:1263-1271 (built in round 4)
    assert m._lora_target_set() == "context" and m.get_lora_target_layers() == ["add_k_proj", "add_q_proj", "add_v_proj", "add_qkv_proj", "to_add_out"]
    # the feed-forward sets and TheLastBen's single-layer sets (flux/model.py:1272-1301, 1363-1375): built, with the reference's own layer lists
    want = {"context+ffs": ["add_k_proj", "add_q_proj", "add_v_proj", "add_qkv_proj", "to_add_out", "ff_context.net.0.proj", "ff_context.net.2"],
            "all+ffs": ["to_k", "to_q", "to_v", "to_qkv", "add_qkv_proj", "add_k_proj", "add_q_proj", "add_v_proj", "to_out.0", "to_add_out",
                        "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2", "proj_mlp", "proj_out"],
            "all+ffs+embedder": ["x_embedder", "to_k", "to_q", "to_v", "to_qkv", "add_qkv_proj", "to_out.0", "add_k_proj", "add_q_proj", "add_v_proj", "to_add_out",
                                 "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2", "proj_mlp", "proj_out"],
            "ai-toolkit": ["to_q", "to_k", "to_qkv", "add_qkv_proj", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out", "ff.net.0.proj", "ff.net.2",
                           "ff_context.net.0.proj", "ff_context.net.2", "norm.linear", "norm1.linear", "norm1_context.linear", "proj_mlp", "proj_out"],
            "tiny": ["single_transformer_blocks.7.proj_out", "single_transformer_blocks.20.proj_out"],
            "nano": ["single_transformer_blocks.7.proj_out"]}
    for v, layers in want.items():
        m.config = SimpleNamespace(flux_lora_target=v)
        assert m._lora_target_set() == v and m.get_lora_target_layers() == layers
    for v in ("controlnet",):                                                       # the layers of a Flux ControlNet (a model this path does not build): refused
        m.config = SimpleNamespace(flux_lora_target=v)
        with pytest.raises(NotImplementedError, match="flux_lora_target"):
            m._lora_target_set()
