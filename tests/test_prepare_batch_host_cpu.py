"""`prepare_batch` host logic on the CPU (noising through tests/ops_emulator.py): the flow-matching branch's `input_perturbation` (common.py:5957-5968 with
`_prepare_flow_noisy_latents`, :4975-4992): the INPUT noise is perturbed, the prediction target keeps the un-perturbed noise; and the known-answer values of
tests/test_prepare_batch_gpu.py (the reference's tests/test_flux_model.py:122-124) for the plain branch."""
from types import SimpleNamespace
from unittest.mock import patch

import torch

from tests import ops_emulator as EMU

BF16 = torch.bfloat16


def _model(monkeypatch, **cfg):
    EMU.install(monkeypatch)
    from simpletuner_amd.foundation import ModelFoundation
    from simpletuner_amd.training.trainer import default_config
    c = default_config(flow_schedule_shift=None, **cfg)
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    m = ModelFoundation(c, acc)
    m.noise_schedule = SimpleNamespace(config=SimpleNamespace(num_train_timesteps=1000))
    return m


def test_plain_flow_values(monkeypatch):
    m = _model(monkeypatch)
    sig = torch.tensor([0.25])
    m.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
    out = m.prepare_batch({"latent_batch": torch.full((1, 4, 8, 8), 2.0, dtype=BF16), "noise": torch.full((1, 4, 8, 8), 6.0, dtype=BF16)}, {"global_step": 0})
    assert torch.allclose(out["noisy_latents"].float(), torch.full((1, 4, 8, 8), 3.0), atol=2e-2)                 # 0.75 * 2 + 0.25 * 6
    assert torch.allclose(m.get_prediction_target(out).float(), torch.full((1, 4, 8, 8), 4.0), atol=2e-2)         # n - x
    assert out["input_noise"] is out["noise"]


def test_input_perturbation_on_the_flow_path_perturbs_the_input_noise_only(monkeypatch):
    m = _model(monkeypatch, input_perturbation=0.5, input_perturbation_steps=100)
    sig = torch.tensor([0.25, 0.5])
    m.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
    lat = torch.full((2, 4, 8, 8), 2.0, dtype=BF16)
    noise = torch.full((2, 4, 8, 8), 6.0, dtype=BF16)
    eps = torch.full((2, 4, 8, 8), 2.0, dtype=BF16)
    with patch("torch.randn_like", return_value=eps):
        out = m.prepare_batch({"latent_batch": lat, "noise": noise}, {"global_step": 50})
    p = 0.5 * (1.0 - 50 / 100)                                   # the strength decays linearly over input_perturbation_steps
    want_in = 6.0 + p * 2.0
    assert torch.allclose(out["input_noise"].float(), torch.full((2, 4, 8, 8), want_in), atol=3e-2)
    assert torch.equal(out["noise"], noise)
    x = out["noisy_latents"].float()
    assert torch.allclose(x[0], torch.full_like(x[0], 0.75 * 2.0 + 0.25 * want_in), atol=3e-2)
    assert torch.allclose(x[1], torch.full_like(x[1], 0.5 * 2.0 + 0.5 * want_in), atol=3e-2)
    assert torch.allclose(m.get_prediction_target(out).float(), torch.full((2, 4, 8, 8), 4.0), atol=2e-2)         # the target: un-perturbed n - x
    # past input_perturbation_steps the option is off again
    with patch("torch.randn_like", return_value=eps):
        out = m.prepare_batch({"latent_batch": lat, "noise": noise}, {"global_step": 100})
    assert out["input_noise"] is out["noise"]
    assert torch.allclose(out["noisy_latents"].float()[0], torch.full((4, 8, 8), 3.0), atol=2e-2)
