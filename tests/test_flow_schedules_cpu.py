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


# ---- cubic-spline sigma density (flow_cubic_schedule_weights) ----
# tests/golden/cubic_schedule_vectors.pt = the reference's CubicSplineDistribution executed as written (tools/gen_golden.py::gen_cubic_schedule);
# the known-answer cases below are the reference's tests/test_flow_cubic_schedule.py:52-108.
def _cubic_golden():
    from pathlib import Path
    return torch.load(Path(__file__).parent / "golden" / "cubic_schedule_vectors.pt", weights_only=False)


def test_cubic_density_tables_and_samples_match_reference_vectors():
    from simpletuner_amd.training.sigma_density import CubicSplineDistribution, pchip_slopes
    G = _cubic_golden()
    assert len(G["cases"]) == 7
    for c in G["cases"]:
        d = CubicSplineDistribution(c["weights"])
        torch.testing.assert_close(pchip_slopes(torch.tensor(c["weights"]), 1.0 / (len(c["weights"]) - 1)), c["slopes"], rtol=0, atol=0)
        torch.testing.assert_close(d.pdf_grid, c["pdf"], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(d.cdf_grid, c["cdf"], rtol=1e-6, atol=1e-7)
        torch.manual_seed(c["seed"])
        torch.testing.assert_close(d.sample((257,)), c["samples"], rtol=1e-6, atol=1e-6)
        lp = d.log_prob(c["query"])
        assert torch.equal(torch.isinf(lp), torch.isinf(c["log_prob"]))
        fin = torch.isfinite(c["log_prob"])
        torch.testing.assert_close(lp[fin], c["log_prob"][fin], rtol=1e-5, atol=1e-5)


def test_cubic_weight_parsing_matches_reference_vectors_and_rejects_invalid():
    import math

    from simpletuner_amd.training.sigma_density import parse_cubic_spline_weights
    for raw, want in _cubic_golden()["parse"]:
        assert parse_cubic_spline_weights(raw) == want, (raw, want)
    for bad in ([1.0, -1.0], [0.0, 0.0], [1.0, math.inf], "1,,2", {"weight": 1}):         # test_flow_cubic_schedule.py:85-89
        with pytest.raises(ValueError):
            parse_cubic_spline_weights(bad)


def test_cubic_empty_and_single_weight_schedules_are_uniform():
    from simpletuner_amd.training.sigma_density import CubicSplineDistribution
    for weights in ([], [0.0], [4.0]):                                                       # test_flow_cubic_schedule.py:53-59
        torch.manual_seed(17)
        want = torch.rand(64)
        torch.manual_seed(17)
        assert torch.equal(CubicSplineDistribution(weights).sample((64,)), want)


def test_cubic_two_weights_define_linear_density_and_bounds():
    import math

    from simpletuner_amd.training.sigma_density import CubicSplineDistribution
    d = CubicSplineDistribution([0.0, 1.0])                                                  # test_flow_cubic_schedule.py:61-68
    torch.testing.assert_close(d.log_prob(torch.tensor([0.25, 0.75])).exp(), torch.tensor([0.5, 1.5]), atol=1e-3, rtol=0)
    torch.manual_seed(11)
    assert abs(d.sample((100_000,)).mean().item() - 2.0 / 3.0) < 5e-3
    d = CubicSplineDistribution([0.0, 1.0, 0.1, 2.0, 0.0])                                  # :70-77
    torch.manual_seed(5)          # the reference draws unseeded; a draw landing exactly on a zero-density knot would make log_prob -inf
    s = d.sample((10_000,))
    assert s.min().item() >= 0.0 and s.max().item() <= 1.0 and torch.isfinite(d.log_prob(s)).all()
    assert d.log_prob(torch.tensor([-0.1, 1.1])).tolist() == [-math.inf, -math.inf]


def test_cubic_schedule_feeds_the_flow_sampler_with_shift():
    from simpletuner_amd.training.sigma_density import CubicSplineDistribution
    m = _model(flow_cubic_schedule_weights=[0.0, 1.0], flow_schedule_shift=2.0, flow_timesteps_mode="fixed-list")   # test_flow_cubic_schedule.py:93-108
    batch = {"latents": torch.zeros(32, 1, 2, 2), "noise": torch.zeros(32, 1, 2, 2)}
    torch.manual_seed(7)
    raw = CubicSplineDistribution([0.0, 1.0]).sample((32,))
    want = (raw * 2.0) / (1.0 + raw)
    torch.manual_seed(7)
    sigmas, timesteps = m.sample_flow_sigmas(batch=batch, state={})
    torch.testing.assert_close(sigmas, want)
    torch.testing.assert_close(timesteps, want * 1000.0)
    # the option wins over the uniform / beta / fast switches and is off for None / "" / "none"
    assert _model(flow_cubic_schedule_weights="none")._uses_flow_cubic_schedule() is False
    assert _model(flow_cubic_schedule_weights=[])._uses_flow_cubic_schedule() is True


# ---- round-robin cursor in checkpoints (reference tests/test_flow_custom_timesteps.py:90-127) ----
def test_round_robin_resume_reset_overrides_prior_cursor():
    m = _rr("100,200,300,400,500")
    m.accelerator.num_processes = 2
    batch = {"latents": torch.zeros(2, 1, 2, 2)}
    m.sample_flow_sigmas(batch, state={"global_step": 0})
    m.reset_flow_custom_timestep_cursor(global_step=1)
    _, t = m.sample_flow_sigmas(batch, state={"global_step": 1})
    assert torch.equal(t, torch.tensor([500.0, 100.0]))


def test_round_robin_checkpoint_restores_microbatch_cursor(tmp_path):
    m = _rr("100,200,300,400")
    batch = {"latents": torch.zeros(1, 1, 2, 2)}
    for _ in range(3):
        m.sample_flow_sigmas(batch, state={"global_step": 0})
    m.save_flow_custom_timestep_state(str(tmp_path))
    assert (tmp_path / "flow_custom_timestep_state.json").exists()
    resumed = _rr("100,200,300,400")
    assert resumed.load_flow_custom_timestep_state(str(tmp_path), fallback_global_step=1) is True
    _, t = resumed.sample_flow_sigmas(batch, state={"global_step": 1})
    assert torch.equal(t, torch.tensor([400.0]))


def test_round_robin_checkpoint_load_falls_back_to_global_step(tmp_path):
    m = _rr("100,200,300,400")
    assert m.load_flow_custom_timestep_state(str(tmp_path), fallback_global_step=1) is False
    _, t = m.sample_flow_sigmas({"latents": torch.zeros(1, 1, 2, 2)}, state={"global_step": 1})
    assert torch.equal(t, torch.tensor([200.0]))


def test_fixed_list_mode_writes_no_cursor_state(tmp_path):
    m = _rr("100,200,300", mode="fixed-list")
    m.sample_flow_sigmas({"latents": torch.zeros(1, 1, 2, 2)}, state={})
    m.save_flow_custom_timestep_state(str(tmp_path))
    assert list(tmp_path.iterdir()) == []


# ---- per-dataset timestep_sampling_offset (reference tests/test_timestep_bias_sampling.py) ----
def test_dataset_timestep_sampling_offset_lookup_and_effect():
    import simpletuner_amd.foundation as F
    calls = []

    def lookup(backend_id):
        calls.append(backend_id)
        return {"ds1": {"timestep_sampling_offset": -0.5}, "ds2": {}}.get(backend_id)

    with patch.object(F, "DATA_BACKEND_CONFIGS", lookup):
        assert ModelFoundation._get_dataset_timestep_sampling_offset(object(), {"data_backend_id": "ds1"}) == -0.5
        assert ModelFoundation._get_dataset_timestep_sampling_offset(object(), {"data_backend_id": "ds2"}) == 0.0
        assert ModelFoundation._get_dataset_timestep_sampling_offset(object(), {}) == 0.0
        assert calls == ["ds1", "ds2", None]
        # the offset moves the mean of the logit-normal draw: sigma = sigmoid(scale * (N + offset))      (common.py:5066-5072)
        m = _model(flow_sigmoid_scale=1.5)
        batch = {"latents": torch.zeros(4, 1, 2, 2), "noise": torch.zeros(4, 1, 2, 2), "data_backend_id": "ds1"}
        torch.manual_seed(3)
        normal = torch.randn(4)
        torch.manual_seed(3)
        sigmas, timesteps = m.sample_flow_sigmas(batch, state={})
        torch.testing.assert_close(sigmas, torch.sigmoid(1.5 * (normal - 0.5)))
        torch.testing.assert_close(timesteps, sigmas * 1000.0)


# ---- dynamic shift mu (reference tests/test_validation_dynamic_shift.py:47-73) ----
def test_calculate_dynamic_shift_mu_uses_patch_size_and_resolution():
    sched = SimpleNamespace(config=SimpleNamespace(base_image_seq_len=256, max_image_seq_len=512, base_shift=0.5, max_shift=1.0))
    m = _model(patch_size=4)
    m.model = None
    mu = m.calculate_dynamic_shift_mu(sched, torch.zeros(1, 4, 8, 8))
    assert mu == pytest.approx(0.5 / (512 - 256) * 4)                         # the reference's expected value (seq_len 4; the line's intercept is 0)
    assert m.calculate_dynamic_shift_mu(sched, None) is None
    # video latents count frames; the trained component's patch size wins over the config's
    m.model = SimpleNamespace(config=SimpleNamespace(patch_size=2))
    assert m.calculate_dynamic_shift_mu(sched, torch.zeros(1, 4, 3, 8, 8)) == pytest.approx(0.5 / 256 * 48)
    bad = SimpleNamespace(config=SimpleNamespace(base_image_seq_len=None, max_image_seq_len=512, base_shift=0.5, max_shift=1.0))
    with pytest.raises(ValueError, match="base_image_seq_len"):
        m.calculate_dynamic_shift_mu(bad, torch.zeros(1, 4, 4, 4))
