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


# ------------------------------------------------------------------------------------------------------------------------
# r03: DDIM (the reference's default validation scheduler of the epsilon / v families) and classifier-free guidance
# ------------------------------------------------------------------------------------------------------------------------
def test_ddim_timestep_spacing_and_alpha_table():
    from simpletuner_amd.foundation import DDPMSchedule
    from simpletuner_amd.sampling import DDIMScheduler
    sc = DDIMScheduler()
    sc.set_timesteps(20)
    assert sc.timesteps.tolist() == [951 - 50 * i for i in range(20)]            # "leading" spacing, steps_offset 1: the SD / SDXL scheduler_config
    tr = DDIMScheduler(timestep_spacing="trailing")
    tr.set_timesteps(4)
    assert tr.timesteps.tolist() == [999, 749, 499, 249]
    # the same alpha-bar table as the training-side DDPM schedule (pinned to reference code: tests/test_unet_cpu.py)
    assert torch.equal(sc.alphas_cumprod, DDPMSchedule().alphas_cumprod)


@pytest.mark.parametrize("ptype", ["epsilon", "v_prediction", "sample"])
def test_ddim_with_the_true_prediction_walks_the_exact_forward_marginals(ptype):
    """exact-recovery property of deterministic DDIM: fed the TRUE prediction for x_t = sqrt(a_t) x0 + sqrt(1 - a_t) eps, every step lands on
    sqrt(a_t') x0 + sqrt(1 - a_t') eps with the same (x0, eps), and the last step returns sqrt(a_0) x0 + sqrt(1 - a_0) eps (set_alpha_to_one=False)"""
    from simpletuner_amd.sampling import DDIMScheduler, ddim_sample
    g = torch.Generator().manual_seed(3)
    x0, eps = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64), torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    sc = DDIMScheduler(prediction_type=ptype)
    sc.set_timesteps(10)
    acp = sc.alphas_cumprod.double()
    marg = lambda a: a.sqrt() * x0 + (1 - a).sqrt() * eps

    def predict(x, t):
        a = acp[int(t[0])]
        torch.testing.assert_close(x, marg(a), rtol=0, atol=1e-5)                   # the walk stays on the marginals
        return {"epsilon": eps, "sample": x0, "v_prediction": a.sqrt() * eps - (1 - a).sqrt() * x0}[ptype]   # get_velocity (common.py:4649-4653)

    out = ddim_sample(predict, marg(acp[int(sc.timesteps[0])]), sc, 10)
    torch.testing.assert_close(out, marg(sc.final_alpha_cumprod.double()), rtol=0, atol=1e-5)


def test_cfg_pair_batching_runs_the_model_once_per_step_on_negative_then_positive():
    """sd3/pipeline.py:1769-1785: one forward on [negative ; positive], noise_pred = uncond + g (text - uncond)"""
    from types import SimpleNamespace
    from simpletuner_amd.foundation import PredictionTypes
    from simpletuner_amd.sampling import cfg_combine, sample_images
    calls = []

    class Plug:
        PREDICTION_TYPE = PredictionTypes.EPSILON
        LATENT_CHANNEL_COUNT = 4
        config = SimpleNamespace()
        accelerator = SimpleNamespace(device=torch.device("cpu"))

        def model_predict(self, batch):
            calls.append({k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in batch.items() if k in ("noisy_latents", "prompt_embeds", "add_text_embeds")})
            ehs = batch["encoder_hidden_states"].float()
            # prediction = latents scaled by the prompt's mean: the two halves of the pair differ only through their conditioning
            return {"model_prediction": batch["noisy_latents"].float() * ehs.mean(dim=(1, 2)).view(-1, 1, 1, 1)}

    pe, ne = torch.full((2, 3, 5), 2.0), torch.full((2, 3, 5), -1.0)
    x0 = torch.randn(2, 4, 4, 4, generator=torch.Generator().manual_seed(0))
    out = sample_images(Plug(), pe, torch.ones(2, 6), 4, 4, num_inference_steps=3, decode=False, guidance_scale=4.0, negative_prompt_embeds=ne,
                        negative_pooled=torch.zeros(2, 6), latents=x0)
    assert len(calls) == 3 and all(c["noisy_latents"] == (4, 4, 4, 4) and c["prompt_embeds"] == (4, 3, 5) and c["add_text_embeds"] == (4, 6) for c in calls)
    # replay by hand: eps = x * (uncond + g (cond - uncond)) with uncond = -1, cond = 2
    from simpletuner_amd.sampling import DDIMScheduler
    sc = DDIMScheduler(timestep_spacing="trailing")                   # validation.py:2889-2892: the trainer's inference_scheduler_timestep_spacing default
    sc.set_timesteps(3)
    x = x0.to(torch.bfloat16)
    for t in sc.timesteps:
        pair = torch.cat([x.float() * -1.0, x.float() * 2.0]).to(torch.bfloat16)
        x = sc.step(cfg_combine(pair, 4.0), t, x, return_dict=False)[0]
    torch.testing.assert_close(out.float(), x.float(), rtol=0, atol=0)
    assert torch.equal(cfg_combine(torch.tensor([[1.0], [3.0]]), 2.0), torch.tensor([[5.0]]))
    # without negative embeddings (or g <= 1) the model sees the plain batch
    calls.clear()
    sample_images(Plug(), pe, torch.ones(2, 6), 4, 4, num_inference_steps=2, decode=False, guidance_scale=1.0, latents=x0)
    assert all(c["noisy_latents"] == (2, 4, 4, 4) for c in calls)
