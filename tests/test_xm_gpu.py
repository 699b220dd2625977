"""XM noise candidates on the HIP path (reference: xm_mixin.py:448-485; tests/test_flux_model.py:132-165, tests/test_sd3_model.py:104-147).
1. loss selection on the fused loss kernel: winners, loss value, and the gradient (zero on losing rows) vs the torch formula;
2. a UNet-LoRA step with K=3 candidates equals, in loss and in every adapter gradient, the plain step on just the winning noises
   (losers carry exactly zero gradient; mean over K*B weighted rows == mean over the B winners).  Tolerances: loss |d| <= 2e-4 relative,
   gradients rel-L2 <= 2e-2 (bf16 kernels at two batch sizes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _plugin(dev, **over):
    from simpletuner_amd.sdxl.model import SDXL
    from simpletuner_amd.training.trainer import St355Accelerator, default_config
    cfg = default_config(model_family="sdxl", model_type="lora", xm_enabled=True, xm_candidate_count=2, **over)
    pl = SDXL(cfg, St355Accelerator(dev))
    pl.setup_training_noise_schedule()
    return pl


def test_xm_loss_selects_winners_and_zeroes_loser_gradients():
    dev = torch.device("cuda", 0)
    pl = _plugin(dev)
    noise = torch.tensor([0.0, 1.0, 2.0, 3.0], device=dev).view(4, 1, 1, 1).expand(4, 1, 2, 8).contiguous().to(BF16)
    pred0 = torch.tensor([5.0, 1.5, 2.25, -4.0], device=dev).view(4, 1, 1, 1).expand(4, 1, 2, 8).contiguous().to(BF16)
    pred = pred0.clone().requires_grad_(True)
    hidden = torch.arange(4 * 3 * 2, dtype=torch.float32, device=dev).reshape(4, 3, 2)
    pb = {"latents": torch.zeros(4, 1, 2, 8, device=dev, dtype=BF16), "noise": noise, "timesteps": torch.tensor([100, 200, 100, 200], device=dev),
          "metadata": [{"id": 0}, {"id": 1}, {"id": 0}, {"id": 1}], "xm_candidate_count": 2, "xm_original_batch_size": 2}
    out = {"model_prediction": pred, "hidden_states_buffer": {"layer_2": hidden.clone()}, "xm_candidate_count": 2}
    loss, logs = pl.loss_with_logs(pb, out)
    loss.backward()
    # per-row losses [25, .25, .0625, 49] -> candidates [[25, .25], [.0625, 49]] -> winners [1, 0]
    assert out["xm_winner_indices"].tolist() == [1, 0]
    assert loss.item() == pytest.approx((0.0625 + 0.25) / 2, rel=1e-5)
    assert logs["xm_loss"] == pytest.approx(loss.item()) and logs["xm_candidate_loss_mean"] == pytest.approx((25 + 0.25 + 0.0625 + 49) / 4, rel=1e-5)
    assert logs["xm_candidate_0_wins"] == 1.0 and logs["xm_candidate_1_wins"] == 1.0
    assert pb["latents"].shape[0] == 2 and pb["metadata"] == [{"id": 0}, {"id": 1}] and "xm_candidate_count" not in pb and "xm_candidate_count" not in out
    assert out["model_prediction"].shape[0] == 2 and torch.equal(out["hidden_states_buffer"]["layer_2"], hidden[[2, 1]])
    g = pred.grad.float()
    assert torch.count_nonzero(g[0]) == 0 and torch.count_nonzero(g[3]) == 0
    want = 2.0 * (pred0.float() - noise.float()) / (16 * 2)               # d/dpred of mean over the 2 winners of the 16-element means
    assert _rel(g[1], want[1]) < 1e-2 and _rel(g[2], want[2]) < 1e-2
    # the reference's own case (test_flux_model.py:132-160): winners predict their targets exactly -> loss 0, winners [1, 0]
    pb2 = {"latents": torch.zeros(4, 1, 2, 8, device=dev, dtype=BF16), "noise": noise, "timesteps": torch.tensor([100, 200, 100, 200], device=dev)}
    out2 = {"model_prediction": torch.tensor([5.0, 1.0, 2.0, -4.0], device=dev).view(4, 1, 1, 1).expand(4, 1, 2, 8).contiguous().to(BF16), "xm_candidate_count": 2}
    l2, _ = pl.loss_with_logs(pb2, out2)
    assert l2.item() == 0.0 and out2["xm_winner_indices"].tolist() == [1, 0]


def test_xm_unet_step_equals_plain_step_on_the_winning_noises():
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    from tests.test_unet_model_gpu import SMALL
    dev = torch.device("cuda", 0)
    K, B = 3, 2
    pl = _plugin(dev, snr_gamma=5.0)
    pl.xm_config.candidate_count = K
    m = UNet2DConditionModel(device=dev, **SMALL)
    m.init_synthetic(6)
    m.add_lora_adapter(rank=8, alpha=8.0, seed=3, init_b_std=0.05)
    pl.model = m
    g = torch.Generator().manual_seed(5)
    raw = {"latent_batch": torch.randn(B, 4, 16, 16, generator=g).to(BF16).to(dev), "prompt_embeds": torch.randn(B, 9, 128, generator=g).to(BF16).to(dev),
           "add_text_embeds": torch.randn(B, 64, generator=g).to(BF16).to(dev),
           "batch_time_ids": torch.tensor([[64.0, 48.0, 0.0, 0.0, 64.0, 48.0]] * B).to(BF16).to(dev), "timesteps": torch.tensor([37, 811])}
    torch.manual_seed(11)
    pb = pl.prepare_batch(dict(raw), {"global_step": 0})
    out = pl.model_predict(pb)
    assert out["xm_candidate_count"] == K and out["model_prediction"].shape[0] == K * B and pb["noise"].shape[0] == K * B
    assert torch.equal(pb["timesteps"], torch.tensor([37, 811] * K, device=dev))
    all_noise = pb["noise"].clone()
    loss, logs = pl.loss_with_logs(pb, out)
    loss.backward()
    who = out["xm_winner_indices"]
    assert pb["noise"].shape[0] == B and sum(logs[f"xm_candidate_{k}_wins"] for k in range(K)) == B
    win_noise = all_noise.view(K, B, *all_noise.shape[1:])[who, torch.arange(B, device=dev)]
    assert torch.equal(pb["noise"], win_noise)
    got = {n: p.grad.clone() for n, p in m.named_parameters() if ".lora_" in n}
    for p in m.parameters():
        p.grad = None
    # the plain step on the winners: same latents / timesteps / conditioning, the winning noise injected
    pl.xm_config.enabled = False
    pb1 = pl.prepare_batch(dict(raw, noise=win_noise), {"global_step": 0})
    out1 = pl.model_predict(pb1)
    loss1, logs1 = pl.loss_with_logs(pb1, out1)
    loss1.backward()
    assert logs1 is None and abs(loss.item() - loss1.item()) < 2e-4 * max(1.0, abs(loss1.item())), (loss.item(), loss1.item())
    worst = 0.0
    for n, p in m.named_parameters():
        if ".lora_" in n:
            worst = max(worst, _rel(got[n], p.grad))
    print(f"[xm] K={K} winners {who.tolist()} loss {loss.item():.5f} vs {loss1.item():.5f}; worst adapter-gradient rel-L2 {worst:.2e}")
    assert worst < 2e-2
