"""GPU parity of the `flux_lora_target` sets beyond the attention projections (flux/model.py:1272-1301 "context+ffs" / "all+ffs", :1363-1375 "tiny" / "nano") on the HIP
path: adapters on ff.net.* / ff_context.net.* of the double blocks, proj_mlp / proj_out of the single blocks and the model's own proj_out, as K-extensions of the
projections' GEMMs (proj_out of a single block: its two K segments are taken, the low-rank term leaves as a gated-residual launch of its own).  Prediction and every
adapter gradient vs autograd on the oracle, whose target sets are pinned to the executed reference (tests/test_ref_models_cpu.py, tests/golden/ref_flux_lora_sets.pt).

Tolerances as tests/test_flux_model_gpu.py for the prediction (rel-L2 <= 2e-2, cosine >= 0.9995) and the adapter gradients (rel-L2 <= 5e-2 per matrix).  The loss here
is the MSE against a RANDOM unit-variance target (about 2.5, not a training loss): |delta| <= 2e-3 relative, the bar of tests/test_flux_host_sequencing_cpu.py — a
prediction error of 6e-3 rel-L2 moves that quantity by up to 1e-3 relative on its own (measured: 2.4744 vs 2.4770 at rank 80)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import parity_utils as PU  # noqa: E402


def _run(which, layers, single, B, lat_h, lat_w, S_txt, rank=16):
    from oracle import flux as OF
    from simpletuner_amd.flux.transformer import FluxTransformer2DModel
    dev = "cuda:0"
    model = FluxTransformer2DModel(device=dev, **PU.small_flux_cfg(layers=layers, single=single))
    model.init_synthetic(seed=11)
    model.add_lora_adapter(rank=rank, alpha=float(rank), targets=which, init_b_std=0.02)
    names = {n.split(".lora_")[0] for n, _ in model.named_parameters() if ".lora_" in n}
    assert names == set(OF.lora_targets(PU.oracle_cfg(model), which))
    g = torch.Generator().manual_seed(5)
    bf = lambda t: t.to(torch.bfloat16)
    latents = bf(torch.randn(B, 16, lat_h, lat_w, generator=g))
    packed = bf(OF.pack_latents(latents.float()))
    prompt, pooled = bf(torch.randn(B, S_txt, 128, generator=g)), bf(torch.randn(B, 64, generator=g))
    t = torch.rand(B, generator=g) * 0.8 + 0.1
    target = bf(torch.randn(packed.shape, generator=g))
    img_ids, txt_ids = OF.prepare_latent_image_ids(lat_h, lat_w), torch.zeros(S_txt, 3)
    guidance = torch.full((B,), 3.5) if model.config.guidance_embeds else None
    out = model(hidden_states=packed.to(dev), encoder_hidden_states=prompt.to(dev), pooled_projections=pooled.to(dev), timestep=t.to(dev), img_ids=img_ids.to(dev),
                txt_ids=txt_ids.to(dev), guidance=None if guidance is None else guidance.to(dev), return_dict=False)[0]
    loss = ((out.float() - target.to(dev).float()) ** 2).mean()
    loss.backward()
    P, lora, scale = PU.oracle_state(model)
    lp = {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    o_out = OF.flux_forward(P, PU.oracle_cfg(model), packed.float(), prompt.float(), pooled.float(), t, img_ids, txt_ids, guidance, lp, scale)
    o_loss = ((o_out - target.float()) ** 2).mean()
    o_loss.backward()
    r = PU.rel_l2(out.detach().cpu(), o_out.detach())
    assert r < 2e-2 and PU.cos_sim(out.detach().cpu(), o_out.detach()) > 0.9995 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item()), (r, loss.item(), o_loss.item())
    worst = (0.0, "")
    for name, p in model.named_parameters():
        if ".lora_" not in name:
            continue
        key, ab = name.split(".lora_")
        ref = lp[key][0 if ab.startswith("A") else 1].grad
        assert p.grad is not None and ref.norm().item() > 0, name
        rg = PU.rel_l2(p.grad.cpu(), ref)
        worst = max(worst, (rg, name))
        assert rg < 5e-2, (name, rg)
    print(f"[parity] flux {which} L{layers}+{single} B{B} rank {rank}: pred rel_l2={r:.3e}, worst adapter gradient rel_l2={worst[0]:.3e} at {worst[1]}")
    return model


# (2, 32, 32, 256): tile-aligned streams — the fused projection epilogue stays in use and the per-stream feed-forward launches are segmented-row problems with a K-extension
@pytest.mark.parametrize("which,layers,single,B,lat_h,lat_w,S_txt,rank", [("all+ffs", 2, 2, 2, 16, 16, 32, 16), ("all+ffs", 1, 1, 2, 32, 32, 256, 16),
                                                                       ("context+ffs", 2, 1, 1, 16, 24, 40, 16), ("all+ffs", 1, 1, 1, 16, 16, 64, 80)])
def test_flux_feed_forward_target_sets_match_oracle(which, layers, single, B, lat_h, lat_w, S_txt, rank):
    _run(which, layers, single, B, lat_h, lat_w, S_txt, rank)


def test_flux_embedder_target_set_matches_oracle():
    _run("all+ffs+embedder", 1, 2, 2, 16, 16, 32)


@pytest.mark.parametrize("layers,single,B,lat,S_txt", [(2, 2, 2, 16, 32), (1, 1, 2, 32, 256)])
def test_flux_ai_toolkit_target_set_matches_oracle(layers, single, B, lat, S_txt):
    _run("ai-toolkit", layers, single, B, lat, lat, S_txt)


def test_flux_nano_target_set_matches_oracle_and_stops_the_backward_at_block_7():
    model = _run("nano", 1, 9, 2, 16, 16, 32)
    assert model._bwd_stop == 8
