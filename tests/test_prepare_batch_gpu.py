"""prepare_batch known-answer values of the reference on the HIP noising kernels: tests/test_mixflow.py:60-92 (interpolation slowed to 0.55 / 0.95,
model time unchanged; gamma 0 -> (1-s)x + s n = 3.0) and tests/test_flux_model.py:122-124 (0.75 x + 0.25 n, target n - x)."""
from types import SimpleNamespace
from unittest.mock import patch

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _model(**cfg):
    from simpletuner_amd.foundation import ModelFoundation
    from simpletuner_amd.training.trainer import St355Accelerator, default_config
    c = default_config(flow_schedule_shift=None, **cfg)
    m = ModelFoundation(c, St355Accelerator(torch.device("cuda", 0)))
    m.noise_schedule = SimpleNamespace(config=SimpleNamespace(num_train_timesteps=1000))
    return m


def test_mixflow_interpolation_values_on_the_kernel():
    dev = torch.device("cuda", 0)
    m = _model(mixflow_enabled=True, mixflow_gamma=0.8)
    sig = torch.tensor([0.25, 0.75], device=dev)
    m.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
    batch = {"latent_batch": torch.zeros(2, 4, 8, 8, device=dev, dtype=BF16), "noise": torch.ones(2, 4, 8, 8, device=dev, dtype=BF16)}
    with patch("torch.rand_like", return_value=torch.tensor([0.5, 1.0], device=dev)):
        out = m.prepare_batch(batch, {"global_step": 0})
    torch.testing.assert_close(out["timesteps"].cpu(), torch.tensor([250.0, 750.0]))
    torch.testing.assert_close(out["sigmas"].flatten().cpu(), torch.tensor([0.25, 0.75]))
    torch.testing.assert_close(out["mixflow_interpolation_sigmas"].cpu(), torch.tensor([0.55, 0.95]))
    x = out["noisy_latents"].float().cpu()
    assert torch.allclose(x[0], torch.full_like(x[0], 0.55), atol=4e-3) and torch.allclose(x[1], torch.full_like(x[1], 0.95), atol=4e-3)


def test_gamma_zero_and_plain_flow_values():
    dev = torch.device("cuda", 0)
    for cfg in (dict(mixflow_enabled=True, mixflow_gamma=0.0), dict()):
        m = _model(**cfg)
        sig = torch.tensor([0.25], device=dev)
        m.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
        batch = {"latent_batch": torch.full((1, 4, 8, 8), 2.0, device=dev, dtype=BF16), "noise": torch.full((1, 4, 8, 8), 6.0, device=dev, dtype=BF16)}
        out = m.prepare_batch(batch, {"global_step": 0})
        assert torch.allclose(out["noisy_latents"].float(), torch.full((1, 4, 8, 8), 3.0, device=dev), atol=2e-2)       # 0.75*2 + 0.25*6
        assert torch.allclose(m.get_prediction_target(out).float(), torch.full((1, 4, 8, 8), 4.0, device=dev), atol=2e-2)   # n - x
