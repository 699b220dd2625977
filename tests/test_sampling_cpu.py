"""Flow-matching Euler schedule (simpletuner_amd/sampling.py) against the scheduler vendored in the reference tree, executed by
tools/gen_golden.py::gen_flow_match_scheduler (tests/golden/flow_match_scheduler_vectors.pt), plus the reference's bounds tests
(tests/test_flow_match_scheduler_bounds.py:14-41) ported."""
from pathlib import Path

import pytest
import torch

from simpletuner_amd.sampling import FlowMatchEulerDiscreteScheduler, fix_flow_match_euler_schedule_bounds, flow_match_euler_sample

G = torch.load(Path(__file__).parent / "golden" / "flow_match_scheduler_vectors.pt", weights_only=False)


def test_unshifted_bounds_form_matches_vendored_scheduler_outputs():
    assert len(G["cases"]) == 5
    for c in G["cases"]:
        sc = FlowMatchEulerDiscreteScheduler(bounds="unshifted", **c["kw"])
        assert torch.equal(sc.sigmas, c["init_sigmas"]) and torch.equal(sc.timesteps, c["init_timesteps"])
        assert sc.sigma_min == c["sigma_min"] and sc.sigma_max == c["sigma_max"]
        sc.set_timesteps(c["steps"], mu=c["mu"])
        assert torch.equal(sc.sigmas, c["sigmas"]) and torch.equal(sc.timesteps, c["timesteps"])
        x = c["traj"][0]
        for i, t in enumerate(sc.timesteps):
            x = sc.step(c["v"][i], t, x, return_dict=False)[0]
            # the vendored copy routes dx through its omega mean-shift, (dx - m) * 1.0 + m at omega = 0: one fp32 rounding away from x + dx
            torch.testing.assert_close(x, c["traj"][i + 1], rtol=0, atol=1e-6)
        sc2 = FlowMatchEulerDiscreteScheduler(bounds="unshifted", **c["kw"])
        sc2.set_timesteps(c["steps"], mu=c["mu"])
        torch.testing.assert_close(sc2.scale_noise(c["sn_sample"], c["sn_t"], c["sn_noise"]), c["sn_out"], rtol=0, atol=0)


def test_upstream_form_plus_fix_equals_unshifted_form():
    """common.py:4530-4536: diffusers' scheduler (bounds from the shifted sigmas) followed by fix_flow_match_euler_schedule_bounds"""
    for c in G["cases"]:
        sc = fix_flow_match_euler_schedule_bounds(FlowMatchEulerDiscreteScheduler(**c["kw"]))
        assert sc.sigma_min == pytest.approx(c["sigma_min"], abs=1e-7) and sc.sigma_max == pytest.approx(c["sigma_max"], abs=1e-7)
        sc.set_timesteps(c["steps"], mu=c["mu"])
        torch.testing.assert_close(sc.sigmas, c["sigmas"], rtol=0, atol=1e-6)


def test_static_shift_duplicate_shift_regression_and_its_fix():
    """tests/test_flow_match_scheduler_bounds.py:14-31"""
    sc = FlowMatchEulerDiscreteScheduler(num_train_timesteps=10, shift=3.0)
    init = sc.sigmas.clone()
    sc.set_timesteps(num_inference_steps=10)
    assert not torch.allclose(sc.sigmas[:-1], init, atol=1e-6) and sc.sigmas[-2].item() > init[-1].item()      # the shift is applied twice
    sc = fix_flow_match_euler_schedule_bounds(FlowMatchEulerDiscreteScheduler(num_train_timesteps=10, shift=3.0))
    init = sc.sigmas.clone()
    sc.set_timesteps(num_inference_steps=10)
    assert sc.sigma_min == pytest.approx(0.1, abs=1e-6) and sc.sigma_max == pytest.approx(1.0, abs=1e-6)
    assert torch.allclose(sc.sigmas[:-1], init, atol=1e-6)
    dyn = FlowMatchEulerDiscreteScheduler(use_dynamic_shifting=True)
    lo, hi = dyn.sigma_min, dyn.sigma_max
    assert fix_flow_match_euler_schedule_bounds(dyn) is dyn and (dyn.sigma_min, dyn.sigma_max) == (lo, hi)       # dynamic shifting is left alone
    with pytest.raises(ValueError, match="mu"):
        dyn.set_timesteps(4)
    with pytest.raises(ValueError, match="integer indices"):
        sc.step(torch.zeros(1), 3, torch.zeros(1))


def test_sampling_loop_integrates_a_known_velocity_field():
    """x(sigma) = (1 - sigma) x0 + sigma n has velocity n - x0 everywhere: Euler from sigma = 1 to 0 lands on x0 exactly (up to fp32 rounding),
    for any schedule; the loop calls predict once per schedule entry with [B] timesteps in scheduler units"""
    torch.manual_seed(0)
    x0, n = torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8)
    sc = fix_flow_match_euler_schedule_bounds(FlowMatchEulerDiscreteScheduler(shift=3.0))
    seen = []

    def predict(x, t):
        seen.append(t.clone())
        return n - x0

    out = flow_match_euler_sample(predict, n.clone(), sc, num_inference_steps=7)
    torch.testing.assert_close(out, x0, rtol=0, atol=1e-5)
    assert len(seen) == 7 and all(t.shape == (2,) for t in seen) and seen[0][0].item() == pytest.approx(1000.0) and seen[-1][0].item() > 0
