"""Model-level GPU parity: the HIP Flux train step (through the plugin surface -> C ABI) vs the CPU oracle on identical
weights, noised latents and timesteps.

Stated tolerances (parity for the network itself is UNPINNED in the reference — no golden tensors — so these
are our own bars): bf16 kernels vs fp32 oracle — prediction rel-L2 <= 2e-2 and cosine >= 0.9995; loss |delta| <= 1e-3 (north
star); LoRA gradients rel-L2 <= 5e-2 and cosine >= 0.999 per adapter matrix; 10-step loss curve |delta| <= 1e-3 with AdamW.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import parity_utils as PU  # noqa: E402


def _build(layers, single, B, lat_h, lat_w, S_txt, rank=16, seed=3, lr=1e-3):
    from simpletuner_amd.flux.model import Flux
    from simpletuner_amd.training.trainer import St355Accelerator, Trainer, default_config

    dev = torch.device("cuda:0")
    cfg = default_config(lora_rank=rank, train_batch_size=B, seed=seed, lora_init_b_std=0.02, learning_rate=lr, flow_schedule_shift=3.0)
    acc = St355Accelerator(dev)
    plugin = Flux(cfg, acc)
    plugin.load_model(**PU.small_flux_cfg(layers=layers, single=single))
    plugin.add_lora_adapter()
    trainer = Trainer(cfg, plugin, acc)
    cpu, devt = PU.make_inputs(B, lat_h, lat_w, S_txt, 128, 64, dev, seed=seed)
    return plugin, trainer, cpu, devt


def _batch(devt):
    return {"latent_batch": devt["latents"], "prompt_embeds": devt["prompt"], "add_text_embeds": devt["pooled"], "noise": devt["noise"]}


# the last case has tile-aligned streams (256 image + 256 text rows, B = 2): the double blocks then run their per-stream projections as segmented-row
# problems over the joint buffers (FluxTransformer2DModel._problems) instead of one problem per sample
@pytest.mark.parametrize("layers,single,B,lat_h,lat_w,S_txt", [(1, 1, 1, 16, 16, 64), (2, 2, 2, 16, 16, 32), (1, 2, 1, 16, 24, 40), (2, 1, 2, 32, 32, 256)])
def test_flux_step_matches_oracle(layers, single, B, lat_h, lat_w, S_txt):
    plugin, trainer, cpu, devt = _build(layers, single, B, lat_h, lat_w, S_txt)
    model = plugin.get_trained_component()
    sig = devt["sigmas"]
    plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)     # identical timesteps on both sides
    P, lora, scale = PU.oracle_state(model)
    prepared = plugin.prepare_batch(_batch(devt), {"global_step": 0})
    # noising / target algebra vs the reference formulas (common.py:4975-4992, 4610-4611)
    s = cpu["sigmas"].view(-1, 1, 1, 1)
    assert PU.rel_l2(prepared["noisy_latents"], (1 - s) * cpu["latents"] + s * cpu["noise"]) < 4e-3
    assert PU.rel_l2(plugin.get_prediction_target(prepared), cpu["noise"] - cpu["latents"]) < 4e-3
    ts_before = prepared["timesteps"].clone()
    out = plugin.model_predict(prepared)
    assert torch.allclose(prepared["timesteps"], ts_before / 1000.0)          # flux/model.py:739-745 (tests/test_flux_model.py:213)
    loss, _ = plugin.loss_with_logs(prepared, out)
    loss.backward()
    o_loss, o_pred, o_grads = PU.oracle_step(P, PU.oracle_cfg(model), lora, scale, cpu)
    r = PU.rel_l2(out["model_prediction"], o_pred); c = PU.cos_sim(out["model_prediction"], o_pred)
    print(f"[parity] flux pred L{layers}+{single} B{B}: rel_l2={r:.3e} cos={c:.6f}  loss hip={loss.item():.6f} oracle={o_loss.item():.6f}")
    assert r < 2e-2 and c > 0.9995
    assert abs(loss.item() - o_loss.item()) < 1e-3 * max(1.0, abs(o_loss.item()))
    worst = (0.0, "")
    for name, p in model.named_parameters():
        if ".lora_" not in name:
            continue
        key = name.split(".lora_")[0]
        ref = o_grads[key][0 if ".lora_A." in name else 1]
        assert p.grad is not None, name
        rg, cg = PU.rel_l2(p.grad, ref), PU.cos_sim(p.grad, ref)
        worst = max(worst, (rg, name))
        assert rg < 5e-2 and cg > 0.999, f"{name}: rel={rg:.3e} cos={cg:.5f}"
    print(f"[parity] flux lora grads: worst rel_l2={worst[0]:.3e} at {worst[1]}")


def test_flux_loss_curve_matches_oracle_adamw():
    """10 optimizer steps, identical noise/timesteps each step: HIP (bf16 compute, fused fp32 AdamW) vs oracle (fp32, torch.optim.AdamW)."""
    plugin, trainer, cpu, devt = _build(1, 1, 2, 16, 16, 32, rank=8, lr=2e-3)
    model = plugin.get_trained_component()
    sig = devt["sigmas"]
    plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
    P, lora, scale = PU.oracle_state(model)
    ocfg = PU.oracle_cfg(model)
    names = sorted(lora)
    params = {k: (torch.nn.Parameter(lora[k][0].clone()), torch.nn.Parameter(lora[k][1].clone())) for k in names}
    opt = torch.optim.AdamW([t for k in names for t in params[k]], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    hip_losses, ora_losses = [], []
    for step in range(10):
        hip_losses.append(trainer.train_step(_batch(devt)).item())
        opt.zero_grad()
        s = cpu["sigmas"].view(-1, 1, 1, 1)
        noisy = ((1 - s) * cpu["latents"] + s * cpu["noise"]).to(torch.bfloat16).float()
        target = (cpu["noise"] - cpu["latents"]).to(torch.bfloat16).float()
        pred = PU.OF.flux_model_predict(P, ocfg, noisy, cpu["prompt"], cpu["pooled"], cpu["sigmas"] * 1000.0, 1.0,
                                        lora={k: params[k] for k in names}, lora_scale=scale)
        l = ((pred - target) ** 2).mean(dim=(1, 2, 3)).mean()
        l.backward(); opt.step()
        ora_losses.append(l.item())
    d = max(abs(a - b) for a, b in zip(hip_losses, ora_losses))
    print("[parity] loss curve hip   :", [round(x, 5) for x in hip_losses])
    print("[parity] loss curve oracle:", [round(x, 5) for x in ora_losses])
    print(f"[parity] max |delta loss| over 10 steps = {d:.3e}")
    assert d < 1e-3 * max(1.0, max(ora_losses))
    assert hip_losses[-1] < hip_losses[0]          # it trains


def test_ema_and_clipping_in_the_loop():
    from simpletuner_amd.training.ema import EMAModel

    plugin, trainer, cpu, devt = _build(1, 1, 1, 16, 16, 32, rank=8, lr=1e-3)
    trainer.config.max_grad_norm = 0.01
    trainer.ema_model = EMAModel(trainer.config, trainer.accelerator, trainer.params, decay=0.9)
    p0 = [p.detach().clone() for p in trainer.params]
    shadow_ref = [p.clone() for p in p0]
    for step in range(1, 4):
        trainer.train_step(_batch(devt))
        d = trainer.ema_model.get_decay(step)
        for s, p in zip(shadow_ref, trainer.params):
            s.sub_((1 - d) * (s - p.detach()))
    assert trainer.last_grad_norm is not None and math.isfinite(trainer.last_grad_norm.item())
    for s, e in zip(shadow_ref, trainer.ema_model.shadow_params):
        assert torch.allclose(s, e, atol=1e-6)     # tests/test_ema.py tolerance (atol 1e-6)
    assert any(not torch.equal(a, b.detach()) for a, b in zip(p0, trainer.params))


# "aligned": 256 image + 256 text rows per sample — the fused QKV projection epilogue, the fused RoPE backward and the segmented per-stream problems are
# the paths that get re-run from the checkpoints (the other cases have unaligned streams and take the separate passes)
@pytest.mark.parametrize("mode,interval,stride,shape", [("layer", None, None, (16, 16, 32)), ("interval2", 2, None, (16, 16, 32)), ("seg2-stride4", 2, 4, (16, 16, 32)),
                                                        ("interval2-aligned", 2, None, (32, 32, 256))])
def test_checkpointed_gradients_equal_direct_gradients(mode, interval, stride, shape):
    """SURVEY.md §8(f)3 / reference tests/test_gradient_checkpointing_backend.py:49-105 (checkpointed == direct, atol 1e-6): the recompute runs the
    same kernels in the same order, so prediction, loss and every adapter gradient are BIT-identical to the run that keeps its activations"""
    def run(ckpt):
        import gc
        gc.collect(); torch.cuda.empty_cache()
        plugin, trainer, cpu, devt = _build(3, 5, 2, *shape, rank=8)
        plugin.config.gradient_checkpointing = ckpt
        plugin.config.gradient_checkpointing_interval, plugin.config.gradient_checkpointing_segment_stride = interval, stride
        plugin.configure_gradient_checkpointing()
        model = plugin.get_trained_component()
        assert model.gradient_checkpointing is ckpt
        sig = devt["sigmas"]
        plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
        prepared = plugin.prepare_batch(_batch(devt), {"global_step": 0})
        torch.cuda.reset_peak_memory_stats()
        out = plugin.model_predict(prepared)
        loss, _ = plugin.loss_with_logs(prepared, out)
        kept = torch.cuda.memory_allocated()
        loss.backward()
        segs = (model._checkpoint_segments(3), model._checkpoint_segments(5))
        return out["model_prediction"].detach().clone(), loss.detach().clone(), [p.grad.detach().clone() for p in trainer.params], kept, segs
    p0, l0, g0, kept0, _ = run(False)
    p1, l1, g1, kept1, segs = run(True)
    assert any(ck for seg in segs for (_, _, ck) in seg)
    assert torch.equal(p0, p1)
    # the scalar loss is reduced in a fixed order (one workgroup per sample, wave partials summed in wave order): identical bits
    assert torch.equal(l0, l1)
    assert len(g0) == len(g1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
    print(f"[ckpt] {mode}: activations held between forward and backward {kept0 / 2**20:.1f} MiB -> {kept1 / 2**20:.1f} MiB, plans {segs}")
    assert kept1 < kept0


@pytest.mark.parametrize("lat,St", [(16, 32), (32, 256)])       # (32, 256): aligned streams -> fused projection epilogue + bias variants of the fused kernels
def test_flux_attention_masked_training_matches_oracle(lat, St):
    """flux_attention_masked_training (flux/model.py:813-823, flux/transformer.py:170-173, 227-242): the reference hands SDPA the FLOAT mask
    (mask > 0).to(dtype) expanded with ones over the image tokens, i.e. an ADDITIVE +1 on every valid key — restated in the oracle as key_bias"""
    plugin, trainer, cpu, devt = _build(1, 2, 2, lat, lat, St)
    Si = (lat // 2) ** 2
    plugin.config.flux_attention_masked_training = True
    model = plugin.get_trained_component()
    sig = devt["sigmas"]
    plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
    mask = torch.ones(2, St); mask[0, (5 * St) // 8:] = 0; mask[1, (9 * St) // 32:] = 0
    b = _batch(devt); b["encoder_attention_mask"] = mask.to("cuda:0")
    prepared = plugin.prepare_batch(b, {"global_step": 0})
    out = plugin.model_predict(prepared)
    loss, _ = plugin.loss_with_logs(prepared, out)
    loss.backward()
    P, lora, scale = PU.oracle_state(model)
    ocfg = PU.oracle_cfg(model)
    kb = torch.ones(2, St + Si); kb[:, :St] = (mask > 0).float()
    s = cpu["sigmas"].view(-1, 1, 1, 1)
    noisy = ((1 - s) * cpu["latents"] + s * cpu["noise"]).to(torch.bfloat16).float()
    target = (cpu["noise"] - cpu["latents"]).to(torch.bfloat16).float()
    lp = {k: (a.clone().requires_grad_(True), bb.clone().requires_grad_(True)) for k, (a, bb) in lora.items()}
    packed = PU.OF.pack_latents(noisy)
    o = PU.OF.flux_forward(P, ocfg, packed, cpu["prompt"], cpu["pooled"], cpu["sigmas"], PU.OF.prepare_latent_image_ids(lat, lat), torch.zeros(St, 3),
                           torch.full((2,), 1.0), lp, scale, key_bias=kb)
    o_pred = PU.OF.unpack_latents(o, lat, lat)
    o_loss = ((o_pred - target) ** 2).mean(dim=(1, 2, 3)).mean()
    o_loss.backward()
    r = PU.rel_l2(out["model_prediction"], o_pred)
    # the mask must matter: the unmasked oracle prediction differs by far more than the parity tolerance
    o_nomask = PU.OF.unpack_latents(PU.OF.flux_forward(P, ocfg, packed, cpu["prompt"], cpu["pooled"], cpu["sigmas"], PU.OF.prepare_latent_image_ids(lat, lat),
                                                       torch.zeros(St, 3), torch.full((2,), 1.0), None, 1.0), lat, lat)
    print(f"[parity] flux masked attention: pred rel_l2={r:.3e}  loss hip={loss.item():.6f} oracle={o_loss.item():.6f}")
    assert r < 2e-2 and abs(loss.item() - o_loss.item()) < 1e-3 * max(1.0, abs(o_loss.item()))
    for name, p in model.named_parameters():
        if ".lora_" in name:
            ref = lp[name.split(".lora_")[0]][0 if ".lora_A." in name else 1].grad
            assert PU.rel_l2(p.grad, ref) < 5e-2, name



@pytest.mark.parametrize("start,end,ckpt", [(1, 2, False), (1, -2, True), (3, 5, False), (4, -1, False), (0, 1, True)])
def test_flux_tread_routing_matches_oracle(start, end, ckpt):
    """TREAD on Flux (helpers/training/tread.py; flux/transformer.py:1101-1133, 1211-1241, 1394-1486): 3 double + 4 single blocks, half of the image tokens routed
    around the blocks [start, end] (global layer indices, negative = from the end) — inside the double stack ending on its LAST block (the joint sequence is
    re-opened), across the double/single boundary, starting on single block 0, ending on the last block, starting on block 0.  The HIP step against the
    oracle replaying the SAME permutations (the oracle's routing is pinned to the executed reference model: tests/test_ref_models_cpu.py tread_double /
    tread_single); with per-block recomputation (what the reference falls back to under routing) the result is bit-identical."""
    from simpletuner_amd.training.tread import ReplayRouter

    def run(with_ckpt):
        plugin, trainer, cpu, devt = _build(3, 4, 2, 16, 16, 32, rank=8)
        model = plugin.get_trained_component()
        g = torch.Generator().manual_seed(17)
        B, Si = 2, 64
        perm = torch.stack([torch.randperm(Si, generator=g) for _ in range(B)])
        K = Si - int(round(Si * 0.5))
        rec = {"mask": torch.ones(B, Si, dtype=torch.bool).scatter_(1, perm[:, :K], False), "ids_keep": perm[:, :K], "ids_mask": perm[:, K:], "ids_shuffle": perm,
               "ids_restore": torch.argsort(perm, dim=1)}
        routes = [{"selection_ratio": 0.5, "start_layer_idx": start, "end_layer_idx": end}]
        model.set_router(ReplayRouter([rec]), routes)
        model.train()
        if with_ckpt:
            model.enable_gradient_checkpointing()
        sig = devt["sigmas"]
        plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
        P, lora, scale = PU.oracle_state(model)
        prepared = plugin.prepare_batch(_batch(devt), {"global_step": 0})
        out = plugin.model_predict(prepared)
        loss, _ = plugin.loss_with_logs(prepared, out)
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if ".lora_" in n}
        return model, cpu, P, lora, scale, out["model_prediction"].detach().clone(), loss.detach().clone(), grads, rec, routes

    model, cpu, P, lora, scale, pred, loss, grads, rec, routes = run(False)
    s = cpu["sigmas"].view(-1, 1, 1, 1)
    noisy = ((1 - s) * cpu["latents"] + s * cpu["noise"]).to(torch.bfloat16).float()
    target = (cpu["noise"] - cpu["latents"]).to(torch.bfloat16).float()
    lp = {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    ocfg = PU.oracle_cfg(model)
    o_pred = PU.OF.flux_model_predict(P, ocfg, noisy, cpu["prompt"], cpu["pooled"], cpu["sigmas"] * 1000.0, 1.0, lora=lp, lora_scale=scale,
                                      tread={"routes": routes, "mask_infos": [rec]})
    o_loss = ((o_pred - target) ** 2).mean(dim=(1, 2, 3)).mean()
    o_loss.backward()
    rr, cc = PU.rel_l2(pred, o_pred), PU.cos_sim(pred, o_pred)
    print(f"[tread flux] route [{start}, {end}]: routed prediction vs oracle (same permutations): rel-L2 {rr:.3e} cos {cc:.6f}; loss hip {loss.item():.6f} oracle {o_loss.item():.6f}")
    assert rr < 2e-2 and cc > 0.9995 and abs(loss.item() - o_loss.item()) < 1e-3 * max(1.0, abs(o_loss.item()))
    plain = PU.OF.flux_model_predict(P, ocfg, noisy, cpu["prompt"], cpu["pooled"], cpu["sigmas"] * 1000.0, 1.0,
                                     lora={k: (a.detach(), b.detach()) for k, (a, b) in lp.items()}, lora_scale=scale)
    assert PU.rel_l2(plain, o_pred) > 2e-2                  # the route is live: the un-routed prediction differs
    worst = (0.0, "")
    for name, g_ in grads.items():
        ref = lp[name.split(".lora_")[0]][0 if ".lora_A." in name else 1].grad
        rg = PU.rel_l2(g_, ref)
        worst = max(worst, (rg, name))
        assert rg < 5e-2, (name, rg)
    print(f"[tread flux] worst adapter gradient rel-L2 {worst[0]:.3e} at {worst[1]}")
    if ckpt:
        _, _, _, _, _, pred_c, _, grads_c, _, _ = run(True)
        assert torch.equal(pred, pred_c) and all(torch.equal(grads[k], grads_c[k]) for k in grads)


@pytest.mark.parametrize("layers,single,masked", [(1, 3, False), (0, 2, False), (1, 2, True), (3, 2, False), (3, 0, True)])
def test_flux_block_c_entry_points_equal_host_sequencing(layers, single, masked, monkeypatch):
    """st355_block_flux_single_fwd / _bwd and st355_block_flux_double_fwd / _bwd (one transformer block forward / backward as ONE C call) issue
    the same launches in the same order on the same operands as sequencing the per-kernel entry points from the host: prediction, loss and every adapter
    gradient bit-identical.  Tile-aligned streams (256 image + 256 text tokens per sample, B = 2: segmented-row operands over the joint buffers) = the production
    form the entry points are built for; (0, 2): no double blocks, so single block 0 runs through the C backward too (no previous gate); (3, *): double blocks
    1 and 2 through the C backward (block 0 keeps the host's frozen-embedder case), the last one writing the joint sequence in place; masked: the key-bias variants."""
    import simpletuner_amd.flux.transformer as FT

    def run(block_abi):
        monkeypatch.setattr(FT, "_BLOCK_ABI", block_abi)
        plugin, trainer, cpu, devt = _build(layers, single, 2, 32, 32, 256, rank=16)
        model = plugin.get_trained_component()
        sig = devt["sigmas"]
        plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
        if masked:
            plugin.config.flux_attention_masked_training = True
        batch = _batch(devt)
        if masked:
            m = torch.ones(2, 256, device="cuda:0"); m[0, 100:] = 0; m[1, 180:] = 0
            batch["encoder_attention_mask"] = m
        prepared = plugin.prepare_batch(batch, {"global_step": 0})
        out = plugin.model_predict(prepared)
        loss, _ = plugin.loss_with_logs(prepared, out)
        loss.backward()
        return out["model_prediction"].detach().clone(), loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if ".lora_" in n}

    p0, l0, g0 = run(False)
    p1, l1, g1 = run(True)
    assert torch.equal(p0, p1) and torch.equal(l0, l1)
    assert set(g0) == set(g1) and all(torch.equal(g0[k], g1[k]) for k in g0)
    assert len(g0) > 0


@pytest.mark.parametrize("layers,single,B,lat,S_txt", [(1, 2, 1, 32, 256), (2, 1, 2, 32, 256), (1, 1, 1, 16, 64),
                                                       (2, 2, 2, 16, 40)])
def test_flux_tokenwise_timesteps_match_oracle(layers, single, B, lat, S_txt):
    """TOKENWISE timesteps [B, S_img] (CREPA self-flow; reference tests/test_flux_model.py:213-241; flux/transformer.py:245-294, 386-412, 1068-1086, 1505) on the HIP
    path: the AdaLN / gated-residual / scale kernels run with ONE modulation row per token (rows_per_batch = 1) — per image token in the double blocks and norm_out,
    [mean x S_txt || per token] along the single blocks' joint sequence, the token mean on the text stream; the fused QKV + RMSNorm + RoPE projection and the fused
    RoPE backward stay in use (256-row streams), the block-level C entry points step aside.  Prediction and LoRA gradients vs autograd on the oracle, whose
    tokenwise branch is pinned to the executed reference class (tests/test_ref_models_cpu.py)."""
    from oracle import flux as OF
    from simpletuner_amd.flux.transformer import FluxTransformer2DModel
    dev = "cuda:0"
    model = FluxTransformer2DModel(device=dev, **PU.small_flux_cfg(layers=layers, single=single))
    model.init_synthetic(seed=11)
    model.add_lora_adapter(rank=16, alpha=16.0, init_b_std=0.02)
    g = torch.Generator().manual_seed(5)
    bf = lambda t: t.to(torch.bfloat16)
    latents = bf(torch.randn(B, 16, lat, lat, generator=g))
    packed = bf(OF.pack_latents(latents.float()))
    Si = packed.shape[1]
    prompt, pooled = bf(torch.randn(B, S_txt, 128, generator=g)), bf(torch.randn(B, 64, generator=g))
    t = torch.rand(B, Si, generator=g) * 0.9 + 0.05
    target = bf(torch.randn(packed.shape, generator=g))
    img_ids, txt_ids = OF.prepare_latent_image_ids(lat, lat), torch.zeros(S_txt, 3)
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
    assert r < 2e-2 and PU.cos_sim(out.detach().cpu(), o_out.detach()) > 0.9995 and abs(loss.item() - o_loss.item()) < 1e-3 * max(1.0, o_loss.item())
    worst = 0.0
    for name, p in model.named_parameters():
        if ".lora_" not in name:
            continue
        key, which = name.split(".lora_")
        ref = lp[key][0 if which.startswith("A") else 1].grad
        rg = PU.rel_l2(p.grad.cpu(), ref)
        worst = max(worst, rg)
        assert rg < 5e-2, (name, rg)
    print(f"[flux tokenwise] L{layers}+{single} B{B} S_img {Si}: pred rel-L2 {r:.3e}, worst adapter gradient rel-L2 {worst:.3e}")
