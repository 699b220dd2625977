"""Full-rank Flux training on the GPU (`model_type == "full"`; the reference's multi-GPU Flux datapoint trains the whole transformer,
documentation/DISTRIBUTED.md:291-298): the gradient of EVERY parameter — Linear weights and biases (TN weight-gradient GEMMs, column sums), the AdaLN
modulation rows and gates (token-axis reductions), the q / k RMSNorm weights, the embedders — vs fp32 autograd on the oracle, through the plugin surface.

Stated tolerances (bf16 kernels + bf16 gradient storage vs the fp32 oracle, as for the SD3 full fine-tune): prediction rel-L2 <= 2e-2, loss |delta| <= 1e-3;
per-tensor gradient rel-L2 <= 6e-2 and cosine >= 0.998 for tensors that carry signal (norm >= 1e-3 of the largest gradient norm); small stays small.
The same host sequencing runs on the CPU against the kernel-contract emulator in tests/test_flux_host_sequencing_cpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flux as OF  # noqa: E402
from tests import parity_utils as PU  # noqa: E402


def _build_full(layers, single, B, lat_h, lat_w, S_txt, seed=5, lr=1e-4):
    from simpletuner_amd.flux.model import Flux
    from simpletuner_amd.training.trainer import St355Accelerator, default_config

    dev = torch.device("cuda:0")
    cfg = default_config(model_type="full", train_batch_size=B, seed=seed, learning_rate=lr, flow_schedule_shift=3.0)
    acc = St355Accelerator(dev)
    plugin = Flux(cfg, acc)
    plugin.load_model(**PU.small_flux_cfg(layers=layers, single=single))
    plugin.enable_full_finetune()
    cpu, devt = PU.make_inputs(B, lat_h, lat_w, S_txt, 128, 64, dev, seed=seed)
    return plugin, cfg, acc, cpu, devt


def _batch(devt):
    return {"latent_batch": devt["latents"], "prompt_embeds": devt["prompt"], "add_text_embeds": devt["pooled"], "noise": devt["noise"]}


def _step(plugin, devt):
    sig = devt["sigmas"]
    plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
    prepared = plugin.prepare_batch(_batch(devt), {"global_step": 0})
    out = plugin.model_predict(prepared)
    loss, _ = plugin.loss_with_logs(prepared, out)
    loss.backward()
    return out["model_prediction"], loss


# the last case has tile-aligned streams (256 image + 256 text rows, B = 2): the per-stream projections run as segmented-row problems over the joint buffers
@pytest.mark.parametrize("layers,single,B,lat_h,lat_w,S_txt", [(1, 1, 1, 16, 16, 64), (2, 2, 2, 16, 24, 40), (2, 1, 2, 32, 32, 256)])
def test_flux_full_rank_gradients_match_oracle(layers, single, B, lat_h, lat_w, S_txt):
    plugin, cfg, acc, cpu, devt = _build_full(layers, single, B, lat_h, lat_w, S_txt)
    model = plugin.get_trained_component()
    assert not hasattr(model, "lora_groups") or not model.lora_groups
    P, _, _ = PU.oracle_state(model)
    pred_hip, loss = _step(plugin, devt)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    s = cpu["sigmas"].view(-1, 1, 1, 1)
    noisy = ((1 - s) * cpu["latents"] + s * cpu["noise"]).to(torch.bfloat16).float()
    target = (cpu["noise"] - cpu["latents"]).to(torch.bfloat16).float()
    pred = OF.flux_model_predict(Pg, PU.oracle_cfg(model), noisy, cpu["prompt"], cpu["pooled"], cpu["sigmas"] * 1000.0, float(getattr(cfg, "flux_guidance_value", 1.0)))
    o_loss = ((pred - target) ** 2).mean(dim=(1, 2, 3)).mean()
    o_loss.backward()
    r = PU.rel_l2(pred_hip, pred)
    print(f"[parity] flux full-rank L{layers}+{single} B{B}: pred rel_l2={r:.3e}  loss hip={loss.item():.6f} oracle={o_loss.item():.6f}")
    assert r < 2e-2 and abs(loss.item() - o_loss.item()) < 1e-3 * max(1.0, abs(o_loss.item()))
    gmax = max(v.grad.norm().item() for v in Pg.values())
    worst, checked = (0.0, ""), 0
    for name, p in model.named_parameters():
        ref = Pg[name].grad
        assert p.grad is not None, name
        assert torch.isfinite(p.grad.float()).all(), name
        if ref.norm().item() < 1e-3 * gmax:
            assert p.grad.float().norm().item() < 3e-3 * gmax, name           # small stays small
            continue
        rg, cg = PU.rel_l2(p.grad, ref), PU.cos_sim(p.grad, ref)
        worst = max(worst, (rg, name)); checked += 1
        assert rg < 6e-2 and cg > 0.998, f"{name}: rel={rg:.3e} cos={cg:.5f} |ref|={ref.norm().item():.3e}"
    print(f"[parity] flux full-rank grads: {checked} tensors checked, worst rel_l2={worst[0]:.3e} at {worst[1]}")
    assert checked > 40


def test_flux_full_rank_trains_with_fused_optimizers():
    """4 steps with the fused bf16-arena AdamW (fp32 moments), then 4 with AdamWBF16 (the examples' default): ONE launch per step over the whole parameter
    arena, the loss decreases, the K-major copies follow the weights"""
    from simpletuner_amd.training.optimizer import St355AdamW, St355AdamWBF16
    plugin, cfg, acc, cpu, devt = _build_full(2, 2, 2, 16, 16, 24, lr=2e-4)
    model = plugin.get_trained_component()
    params = model.trainable_parameters()
    for Opt, kw in ((St355AdamW, dict(lr=2e-4, weight_decay=1e-2)), (St355AdamWBF16, dict(lr=2e-4, weight_decay=1e-2))):
        opt = Opt(params, **kw)
        losses = []
        for _ in range(4):
            _, loss = _step(plugin, devt)
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(loss.item())
        print(f"[flux full-rank] {Opt.__name__}: losses {[round(x, 5) for x in losses]}")
        assert losses[-1] < losses[0]
        if Opt is St355AdamWBF16:
            assert opt._launches == 4
    blk = model.single[0]
    model._refresh_transposed()
    assert torch.equal(blk.proj_out.wT, blk.proj_out.w.t().contiguous())


def test_flux_full_rank_checkpointed_gradients_equal_direct_gradients():
    """a checkpointed segment is re-run from its kept input with the same kernels in the same order: prediction and every gradient are BIT-identical"""
    def run(ckpt):
        plugin, cfg, acc, cpu, devt = _build_full(2, 3, 2, 16, 16, 32)
        model = plugin.get_trained_component()
        if ckpt:
            model.enable_gradient_checkpointing()
            model.set_gradient_checkpointing_interval(2)
        pred, _ = _step(plugin, devt)
        return pred, {n: p.grad.clone() for n, p in model.named_parameters()}

    p0, g0 = run(False)
    p1, g1 = run(True)
    assert torch.equal(p0, p1)
    bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    assert not bad, bad[:5]
