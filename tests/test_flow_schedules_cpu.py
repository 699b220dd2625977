"""Known-answer tests of the reference for flow-matching sigma / timestep sampling, ported onto the drop-in's ModelFoundation:
tests/test_mixflow.py:45-104 (sigma = 1 - sqrt(U); interpolation slowed by gamma, model time unchanged; gamma bounds) and
tests/test_flow_custom_timesteps.py:32-90 (round-robin cursor, rank offsets, rank-varying batch sizes, resume step, invalid mode)."""
from types import SimpleNamespace
from unittest.mock import patch

import pytest
import torch

from simpletuner_amd.foundation import ModelFoundation, apply_flow_schedule_shift


def _model(**cfg):
    base = dict(mixflow_enabled=False, mixflow_gamma=0.8, flow_schedule_shift=None, flow_schedule_auto_shift=False, flow_custom_timesteps=None,
                flux_fast_schedule=False, flow_use_beta_schedule=False, flow_use_uniform_schedule=False, flow_sigmoid_scale=1.0)
    base.update(cfg)
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0)
    m = ModelFoundation(SimpleNamespace(**base), acc)
    m.noise_schedule = SimpleNamespace(config=SimpleNamespace(num_train_timesteps=1000))
    return m


def test_mixflow_samples_beta_one_two_noise_sigmas():
    m = _model(mixflow_enabled=True)
    batch = {"latents": torch.zeros(3, 1, 1, 1), "noise": torch.zeros(3, 1, 1, 1)}
    with patch("torch.rand", return_value=torch.tensor([0.0, 0.25, 1.0])):
        sigmas, timesteps = m.sample_flow_sigmas(batch, state={})
    torch.testing.assert_close(sigmas, torch.tensor([1.0, 0.5, 0.0]))
    torch.testing.assert_close(timesteps, torch.tensor([1000.0, 500.0, 0.0]))


def test_mixflow_slows_interpolation_without_changing_model_time():
    m = _model(mixflow_enabled=True, mixflow_gamma=0.8)
    inter = m._mixflow_interpolation_sigmas(torch.tensor([0.25, 0.75]), torch.tensor([0.5, 1.0]))
    torch.testing.assert_close(inter, torch.tensor([0.55, 0.95]))                      # test_mixflow.py:72-73
    m0 = _model(mixflow_enabled=True, mixflow_gamma=0.0)
    torch.testing.assert_close(m0._mixflow_interpolation_sigmas(torch.tensor([0.25])), torch.tensor([0.25]))   # gamma 0: 0.75*2 + 0.25*6 = 3.0 upstream
    with pytest.raises(ValueError, match="mixflow_gamma"):
        _model(mixflow_enabled=True, mixflow_gamma=1.1)._mixflow_interpolation_sigmas(torch.tensor([0.5]))


def test_mixflow_applies_schedule_shift_before_slowing():
    m = _model(mixflow_enabled=True, flow_schedule_shift=3.0)
    batch = {"latents": torch.zeros(1, 1, 1, 1), "noise": torch.zeros(1, 1, 1, 1)}
    with patch("torch.rand", return_value=torch.tensor([0.25])):
        sigmas, _ = m.sample_flow_sigmas(batch, state={})
    torch.testing.assert_close(sigmas, apply_flow_schedule_shift(m.config, m.noise_schedule, torch.tensor([0.5]), batch["noise"]))


def _rr(spec, mode="round-robin"):
    return _model(flow_custom_timesteps=spec, flow_timesteps_mode=mode)


def test_round_robin_cycles_custom_timesteps():
    m = _rr("100,200,300")
    batch = {"latents": torch.zeros(2, 1, 2, 2)}
    _, t1 = m.sample_flow_sigmas(batch, state={})
    _, t2 = m.sample_flow_sigmas(batch, state={})
    assert torch.equal(t1, torch.tensor([100.0, 200.0])) and torch.equal(t2, torch.tensor([300.0, 100.0]))


def test_invalid_custom_timestep_mode_raises():
    with pytest.raises(ValueError, match="flow_timesteps_mode"):
        _rr("100,200", "sequential").sample_flow_sigmas({"latents": torch.zeros(1, 1, 2, 2)}, state={})


def test_round_robin_offsets_distributed_ranks_and_varying_batch_sizes():
    r0, r1 = _rr("100,200,300,400,500"), _rr("100,200,300,400,500")
    for m in (r0, r1):
        m.accelerator.num_processes = 2
    r1.accelerator.process_index = 1
    batch = {"latents": torch.zeros(2, 1, 2, 2)}
    _, a = r0.sample_flow_sigmas(batch, state={"global_step": 0})
    _, b = r1.sample_flow_sigmas(batch, state={"global_step": 0})
    _, c = r0.sample_flow_sigmas(batch, state={"global_step": 0})
    assert torch.equal(a, torch.tensor([100.0, 200.0])) and torch.equal(b, torch.tensor([300.0, 400.0])) and torch.equal(c, torch.tensor([500.0, 100.0]))
    r0, r1 = _rr("100,200,300,400,500,600,700,800"), _rr("100,200,300,400,500,600,700,800")
    for m in (r0, r1):
        m.accelerator.num_processes = 2
        m.accelerator.gather = lambda _t: torch.tensor([1, 3])
    r1.accelerator.process_index = 1
    _, a = r0.sample_flow_sigmas({"latents": torch.zeros(1, 1, 2, 2)}, state={"global_step": 0})
    _, b = r1.sample_flow_sigmas({"latents": torch.zeros(3, 1, 2, 2)}, state={"global_step": 0})
    _, c = r0.sample_flow_sigmas({"latents": torch.zeros(1, 1, 2, 2)}, state={"global_step": 0})
    assert torch.equal(a, torch.tensor([100.0])) and torch.equal(b, torch.tensor([200.0, 300.0, 400.0])) and torch.equal(c, torch.tensor([500.0]))


def test_round_robin_resume_step_and_reset():
    m = _rr("100,200,300,400,500")
    m.accelerator.num_processes = 2
    batch = {"latents": torch.zeros(2, 1, 2, 2)}
    _, t = m.sample_flow_sigmas(batch, state={"global_step": 1})
    assert torch.equal(t, torch.tensor([500.0, 100.0]))
    m2 = _rr("100,200,300,400,500")
    m2.accelerator.num_processes = 2
    m2.sample_flow_sigmas(batch, state={"global_step": 0})
    m2.reset_flow_custom_timestep_cursor(global_step=1)
    _, t = m2.sample_flow_sigmas(batch, state={"global_step": 1})
    assert torch.equal(t, torch.tensor([500.0, 100.0]))


def test_custom_sigma_list_single_value_and_fast_schedule():
    m = _model(flow_custom_timesteps="0.25")
    s, t = m.sample_flow_sigmas({"latents": torch.zeros(3, 1, 2, 2)}, state={})
    assert torch.equal(s, torch.full((3,), 0.25)) and torch.equal(t, torch.full((3,), 250.0))
    m = _model(flux_fast_schedule=True)
    s, t = m.sample_flow_sigmas({"latents": torch.zeros(16, 1, 2, 2)}, state={})
    assert set(s.tolist()) <= {1.0, 0.75, 0.5, 0.25} and torch.equal(t, s * 1000.0)
