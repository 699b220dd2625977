"""The SD3 / SD3.5 engine's HOST SEQUENCING on the CPU (see tests/test_flux_host_sequencing_cpu.py and tests/ops_emulator.py): forward + hand-written backward
of sd3/transformer.py against the kernel-contract emulator, compared with autograd on the oracle — the LoRA path (pins the emulator on a path the GPU tests
already prove) and the full fine-tune of BASELINE.json configs[3] (every parameter's gradient, SD3.5 dual attention + q/k RMSNorm weights included)."""
import pytest
import torch

from oracle import sd3 as OS
from tests import ops_emulator as EMU
from tests import parity_utils as PU

BF16 = torch.bfloat16


def _arch(layers, heads=2, head_dim=64, joint_dim=128, pooled=64, qk_norm=None, dual=()):
    return dict(sample_size=32, num_layers=layers, num_attention_heads=heads, attention_head_dim=head_dim, joint_attention_dim=joint_dim,
                caption_projection_dim=heads * head_dim, pooled_projection_dim=pooled, pos_embed_max_size=24, qk_norm=qk_norm, dual_attention_layers=tuple(dual))


def _ocfg(model):
    c = model.config
    return OS.SD3Config(sample_size=c.sample_size, num_layers=c.num_layers, attention_head_dim=c.attention_head_dim,
                        num_attention_heads=c.num_attention_heads, joint_attention_dim=c.joint_attention_dim,
                        pooled_projection_dim=c.pooled_projection_dim, pos_embed_max_size=c.pos_embed_max_size, qk_norm=c.qk_norm,
                        dual_attention_layers=tuple(c.dual_attention_layers))


def _model(monkeypatch, layers, sd35=False, seed=11):
    EMU.install(monkeypatch)
    from simpletuner_amd.sd3 import transformer as T
    model = T.SD3Transformer2DModel(device="cpu", **_arch(layers, **(dict(qk_norm="rms_norm", dual=(0, 1)) if sd35 else {})))
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".norm_q." in name or ".norm_k." in name or ".norm_added_" in name:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / (p[0].numel() ** 0.5))
        c = model.config
        model.pos_embed.pos_embed.copy_(T.sincos_2d(model.D, c.pos_embed_max_size, c.sample_size // c.patch_size)[None])
    return model


def _inputs(B, lat_h, lat_w, S_txt, seed=5):
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(BF16)
    return dict(lat=bf(torch.randn(B, 16, lat_h, lat_w, generator=g)), prompt=bf(torch.randn(B, S_txt, 128, generator=g)), pooled=bf(torch.randn(B, 64, generator=g)),
                t=(torch.rand(B, generator=g) * 0.8 + 0.1) * 1000.0, target=bf(torch.randn(B, 16, lat_h, lat_w, generator=g)))


def _hip_side(model, d):
    out = model(hidden_states=d["lat"], encoder_hidden_states=d["prompt"], pooled_projections=d["pooled"], timestep=d["t"], return_dict=False)[0]
    loss = ((out.float() - d["target"].float()) ** 2).mean()
    loss.backward()
    return out.detach(), loss.detach()


def _oracle_side(model, d, params_need_grad, lora=None, lora_scale=1.0):
    P, _, _ = PU.oracle_state(model)
    P = {k: (v.clone().requires_grad_(True) if params_need_grad else v) for k, v in P.items()}
    P["pos_embed.pos_embed"] = model.pos_embed.pos_embed.detach().float()
    lp = None if lora is None else {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    out = OS.sd3_forward(P, _ocfg(model), d["lat"].float(), d["prompt"].float(), d["pooled"].float(), d["t"], lora=lp, lora_scale=lora_scale)
    loss = ((out - d["target"].float()) ** 2).mean()
    loss.backward()
    return out.detach(), loss.detach(), P, lp


@pytest.mark.parametrize("sd35", [False, True])
def test_lora_path_through_the_emulator_matches_the_oracle(monkeypatch, sd35):
    model = _model(monkeypatch, 3, sd35)
    model.add_lora_adapter(rank=16, alpha=16.0, init_b_std=0.02)
    d = _inputs(2, 16, 24, 33)
    EMU.BLOCK_CALLS.clear()
    out, loss = _hip_side(model, d)
    # the single-attention blocks ran through the block-level entry points (their emulation restates csrc/blocks.hip), SD3.5's dual-attention blocks through
    # the host-side sequencing
    n_c = sum(1 for b in model.blocks if not b.dual)
    assert EMU.BLOCK_CALLS.get("sd3_fwd", 0) == n_c and EMU.BLOCK_CALLS.get("sd3_bwd", 0) == n_c, (EMU.BLOCK_CALLS, n_c)
    _, lora, scale = PU.oracle_state(model)
    o_out, o_loss, _, lp = _oracle_side(model, d, False, lora, scale)
    assert PU.rel_l2(out, o_out) < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item())
    worst = 0.0
    for name, p in model.named_parameters():
        if ".lora_" not in name:
            continue
        key, which = name.split(".lora_")
        ref = lp[key][0 if which.startswith("A") else 1].grad
        worst = max(worst, PU.rel_l2(p.grad, ref))
        assert PU.rel_l2(p.grad, ref) < 5e-2, name
    print(f"[emu] sd3{'.5' if sd35 else ''} LoRA host sequencing: pred rel_l2={PU.rel_l2(out, o_out):.3e}, worst adapter gradient rel_l2={worst:.3e}")


@pytest.mark.parametrize("layers,B,lat_h,lat_w,S_txt,sd35", [(2, 1, 16, 16, 40, False), (3, 2, 16, 24, 33, False), (3, 2, 16, 24, 33, True)])
def test_full_finetune_gradients_of_every_parameter_match_the_oracle(monkeypatch, layers, B, lat_h, lat_w, S_txt, sd35):
    model = _model(monkeypatch, layers, sd35)
    model.enable_full_finetune()
    d = _inputs(B, lat_h, lat_w, S_txt)
    EMU.BLOCK_CALLS.clear()
    out, loss = _hip_side(model, d)
    n_c = sum(1 for b in model.blocks if not b.dual)
    assert EMU.BLOCK_CALLS.get("sd3_fwd", 0) == n_c and EMU.BLOCK_CALLS.get("sd3_bwd", 0) == (0 if sd35 else n_c), (EMU.BLOCK_CALLS, n_c)
    o_out, o_loss, P, _ = _oracle_side(model, d, True)
    r = PU.rel_l2(out, o_out)
    assert r < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item())
    gmax = max(v.grad.norm().item() for k, v in P.items() if k != "pos_embed.pos_embed")
    worst, checked = (0.0, ""), 0
    for name, p in model.named_parameters():
        ref = P[name].grad
        assert p.grad is not None, name
        if ref.norm().item() < 1e-3 * gmax:
            assert p.grad.float().norm().item() < 3e-3 * gmax, name
            continue
        rg, cg = PU.rel_l2(p.grad, ref), PU.cos_sim(p.grad, ref)
        worst = max(worst, (rg, name)); checked += 1
        assert rg < 6e-2 and cg > 0.998, f"{name}: rel={rg:.3e} cos={cg:.5f} |ref|={ref.norm().item():.3e}"
    print(f"[emu] sd3{'.5' if sd35 else ''} full fine-tune host sequencing L{layers} B{B}: pred rel_l2={r:.3e}; {checked} tensors, worst gradient rel_l2={worst[0]:.3e} at {worst[1]}")
    assert checked > 20


def test_modulation_scale_gradient_survives_a_scale_entry_of_exactly_minus_one(monkeypatch):
    """d scale = sum_t dY * LN(x) is computed from LN(x) itself.  The earlier form recovered LN(x) from the saved modulated output as (n - shift) / (1 + scale),
    which is singular where a modulation scale equals -1 in bf16 — a value random-init tables produce in a few entries per step.  Force one and check."""
    model = _model(monkeypatch, 2)
    model.enable_full_finetune()
    with torch.no_grad():            # bias of block 0's norm1 scale_msa chunk, channel 5: large negative weight row zeroed, bias -1 => scale == -1 for every sample
        blk = model.blocks[0]
        D = model.D
        model.mod_w[blk.mod_off + D + 5].zero_()
        model.mod_b[blk.mod_off + D + 5] = -1.0
    d = _inputs(2, 16, 16, 24)
    _hip_side(model, d)
    _, _, P, _ = _oracle_side(model, d, True)
    name = "transformer_blocks.0.norm1.linear.bias"
    got, ref = dict(model.named_parameters())[name].grad, P[name].grad
    assert torch.isfinite(got.float()).all()
    assert PU.rel_l2(got, ref) < 6e-2, PU.rel_l2(got, ref)


@pytest.mark.parametrize("full", [False, True])
def test_tread_routing_and_checkpoint_plans_through_the_emulator(monkeypatch, full):
    """TREAD on SD3 (sd3/transformer.py:694-706, 796-803): 4 blocks, half of the image tokens routed around blocks [1, -2]; LoRA adapter gradients (or, full fine-tune, every
    parameter's gradient) against the oracle replaying the same permutation; the per-block recompute the reference falls back to under routing is bit-identical; and a
    segmented checkpoint plan (interval 2 / stride 3) without routing is bit-identical to keeping every activation"""
    from simpletuner_amd.training.tread import ReplayRouter
    d = _inputs(2, 16, 16, 24)
    B, Si = 2, 64
    g = torch.Generator().manual_seed(11)
    perm = torch.stack([torch.randperm(Si, generator=g) for _ in range(B)])
    K = Si - int(round(Si * 0.5))
    rec = {"mask": torch.ones(B, Si, dtype=torch.bool).scatter_(1, perm[:, :K], False), "ids_keep": perm[:, :K], "ids_mask": perm[:, K:], "ids_shuffle": perm,
           "ids_restore": torch.argsort(perm, dim=1)}
    routes = [{"selection_ratio": 0.5, "start_layer_idx": 1, "end_layer_idx": -2}]

    def run(route, ckpt, interval=None, stride=None):
        model = _model(monkeypatch, 4)
        if full:
            model.enable_full_finetune()
        else:
            model.add_lora_adapter(rank=8, alpha=8.0, init_b_std=0.02)
        if route:
            model.set_router(ReplayRouter([rec]), routes)
        model.train()
        if ckpt:
            model.gradient_checkpointing = True
            model.gradient_checkpointing_interval, model.gradient_checkpointing_segment_stride = interval, stride
        out, loss = _hip_side(model, d)
        return model, out, loss, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    model, out, loss, grads = run(True, False)
    P, lora, scale = PU.oracle_state(model)
    P = {k: (v.clone().requires_grad_(True) if full else v) for k, v in P.items()}
    P["pos_embed.pos_embed"] = model.pos_embed.pos_embed.detach().float()
    lp = None if full else {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    o_out = OS.sd3_forward(P, _ocfg(model), d["lat"].float(), d["prompt"].float(), d["pooled"].float(), d["t"], lora=lp, lora_scale=scale,
                           tread={"routes": routes, "mask_infos": [rec]})
    o_loss = ((o_out - d["target"].float()) ** 2).mean()
    o_loss.backward()
    assert PU.rel_l2(out, o_out) < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item())
    gmax = max(v.grad.norm().item() for k, v in P.items() if k != "pos_embed.pos_embed") if full else None
    for name, g_ in grads.items():
        if full:
            ref = P[name].grad
            if ref.norm().item() < 1e-3 * gmax:
                continue
            assert PU.rel_l2(g_, ref) < 6e-2, (name, PU.rel_l2(g_, ref))
        else:
            ref = lp[name.split(".lora_")[0]][0 if ".lora_A." in name else 1].grad
            assert PU.rel_l2(g_, ref) < 5e-2, (name, PU.rel_l2(g_, ref))
    _, out_c, _, grads_c = run(True, True)
    assert torch.equal(out, out_c) and all(torch.equal(grads[k], grads_c[k]) for k in grads)
    _, out_p, _, grads_p = run(False, False)
    _, out_s, _, grads_s = run(False, True, 2, 3)
    assert torch.equal(out_p, out_s) and all(torch.equal(grads_p[k], grads_s[k]) for k in grads_p)
    assert not torch.equal(out, out_p)                         # the route is live


def test_cfg_sampling_trajectory_through_the_emulator_matches_the_oracle_driven_loop(monkeypatch):
    """§8(f)4: `sample_images` with classifier-free guidance (one forward per step on [negative ; positive], sd3/pipeline.py:1769-1785) through the SD3 plugin on the CPU
    against the same 4-step Euler loop driven by the oracle forward"""
    from types import SimpleNamespace
    EMU.install(monkeypatch)
    from simpletuner_amd.sampling import FlowMatchEulerDiscreteScheduler, cfg_combine, sample_images
    from simpletuner_amd.sd3.model import SD3
    from simpletuner_amd.training.trainer import default_config
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    plugin = SD3(default_config(model_family="sd3", lora_rank=8, train_batch_size=2, seed=3, lora_init_b_std=0.02, flow_schedule_shift=3.0), acc)
    plugin.load_model(**_arch(2))
    plugin.add_lora_adapter()
    model = plugin.get_trained_component()
    P, lora, scale = PU.oracle_state(model)
    P["pos_embed.pos_embed"] = model.pos_embed.pos_embed.detach().float()
    g = torch.Generator().manual_seed(21)
    bf = lambda t: t.to(BF16)
    pos_p, pos_pool = bf(torch.randn(2, 20, 128, generator=g)), bf(torch.randn(2, 64, generator=g))
    neg_p, neg_pool = bf(torch.randn(2, 20, 128, generator=g)), bf(torch.randn(2, 64, generator=g))
    x0 = bf(torch.randn(2, 16, 16, 16, generator=g))
    gs = 3.5
    with torch.no_grad():
        out = sample_images(plugin, pos_p, pos_pool, 16, 16, num_inference_steps=4, decode=False, guidance_scale=gs, negative_prompt_embeds=neg_p, negative_pooled=neg_pool,
                            latents=x0.clone(), scheduler=FlowMatchEulerDiscreteScheduler(shift=3.0, bounds="unshifted"))
    sc = FlowMatchEulerDiscreteScheduler(shift=3.0, bounds="unshifted")
    sc.set_timesteps(4)
    x = x0.float()
    pe, pp = torch.cat([neg_p.float(), pos_p.float()], 0), torch.cat([neg_pool.float(), pos_pool.float()], 0)
    with torch.no_grad():
        for i, t in enumerate(sc.timesteps):
            pred = OS.sd3_forward(P, _ocfg(model), torch.cat([x, x], 0), pe, pp, t.expand(4), lora=lora, lora_scale=scale)
            x = x + (sc.sigmas[i + 1] - sc.sigmas[i]) * cfg_combine(pred, gs)
    assert torch.isfinite(out.float()).all() and PU.rel_l2(out, x) < 3e-2


@pytest.mark.parametrize("B,lat_h,lat_w", [(1, 16, 24), (2, 32, 32), (2, 16, 24)])      # (2, 16, 24): 96 image rows per sample — per-sample problems slice the per-token gate rows
def test_tokenwise_timesteps_through_the_emulator_match_the_oracle(monkeypatch, B, lat_h, lat_w):
    """TOKENWISE timesteps [B, S_img] (CREPA self-flow; the reference's tests/test_sd3_model.py:179-204 hands them to the transformer; oracle branch pinned to the
    executed reference by tests/test_ref_models_cpu.py): per-token AdaLN rows on the image stream and in norm_out (rows_per_batch = 1), the token mean on the
    context stream; prediction and LoRA gradients against autograd on the oracle.  B = 2 with 256 image rows per sample: segmented problems over per-token gates."""
    model = _model(monkeypatch, 3)
    model.add_lora_adapter(rank=16, alpha=16.0, init_b_std=0.02)
    d = _inputs(B, lat_h, lat_w, 33)
    Si = (lat_h // 2) * (lat_w // 2)
    d["t"] = torch.rand(B, Si, generator=torch.Generator().manual_seed(8)) * 900.0 + 50.0
    out, loss = _hip_side(model, d)
    _, lora, scale = PU.oracle_state(model)
    o_out, o_loss, _, lp = _oracle_side(model, d, False, lora, scale)
    assert PU.rel_l2(out, o_out) < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item())
    # ... and it is not the batch-wise forward in disguise: conditioning every token on its sample's mean timestep moves the prediction
    d_flat = dict(d, t=d["t"].mean(dim=1))
    with torch.no_grad():
        out_flat = model(hidden_states=d_flat["lat"], encoder_hidden_states=d_flat["prompt"], pooled_projections=d_flat["pooled"], timestep=d_flat["t"], return_dict=False)[0]
    assert PU.rel_l2(out_flat, o_out) > 5e-2
    worst = 0.0
    for name, p in model.named_parameters():
        if ".lora_" not in name:
            continue
        key, which = name.split(".lora_")
        ref = lp[key][0 if which.startswith("A") else 1].grad
        worst = max(worst, PU.rel_l2(p.grad, ref))
        assert PU.rel_l2(p.grad, ref) < 5e-2, name
    print(f"[emu] sd3 tokenwise timesteps B{B}: pred rel_l2={PU.rel_l2(out, o_out):.3e}, worst adapter gradient rel_l2={worst:.3e}")


def test_tokenwise_timesteps_refusals(monkeypatch):
    model = _model(monkeypatch, 2)
    d = _inputs(2, 16, 24, 33)
    with torch.no_grad(), pytest.raises(ValueError, match="expected sequence length"):
        model(hidden_states=d["lat"], encoder_hidden_states=d["prompt"], pooled_projections=d["pooled"], timestep=torch.rand(2, 7) * 1000, return_dict=False)
