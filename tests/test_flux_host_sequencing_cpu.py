"""The Flux engine's HOST SEQUENCING on the CPU: forward + hand-written backward of flux/transformer.py executed against tests/ops_emulator.py (plain-torch
stand-ins that honour the kernels' argument contracts), compared with autograd on the oracle.  Which kernel runs on which view, what is saved, which gradient
lands where — all of that is host code and is checked here without a GPU; the kernels themselves are checked by the `-m gpu` parity tests.

  * LoRA path (known-good on the GPU): pins the emulator itself;
  * full-rank training: every weight / bias / q-k RMSNorm weight / modulation row gradient vs the oracle, with and without activation checkpointing."""
import pytest
import torch

from oracle import flux as OF
from tests import ops_emulator as EMU
from tests import parity_utils as PU

BF16 = torch.bfloat16


def _model(monkeypatch, layers, single, guidance=True, head_dim=128, seed=11):
    EMU.install(monkeypatch)
    from simpletuner_amd.flux import transformer as T
    monkeypatch.setattr(T, "_FUSED_QKV", False)               # the fused projection epilogue / block entry points are GPU-only fast paths
    monkeypatch.setattr(T, "_BLOCK_ABI", False)
    model = T.FluxTransformer2DModel(device="cpu", **PU.small_flux_cfg(layers=layers, single=single, head_dim=head_dim, guidance=guidance))
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "norm_q" in name or "norm_k" in name or "norm_added" in name:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / (p.shape[1] ** 0.5))
    return model


def _inputs(B, lat_h, lat_w, S_txt, seed=5):
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(BF16)
    lat = bf(torch.randn(B, 16, lat_h, lat_w, generator=g))
    packed = OF.pack_latents(lat.float())
    d = dict(packed=bf(packed), prompt=bf(torch.randn(B, S_txt, 128, generator=g)), pooled=bf(torch.randn(B, 64, generator=g)),
             t=torch.rand(B, generator=g) * 0.8 + 0.1, target=bf(torch.randn(packed.shape, generator=g)),
             img_ids=OF.prepare_latent_image_ids(lat_h, lat_w), txt_ids=torch.zeros(S_txt, 3), guidance=torch.full((B,), 3.5))
    return d


def _hip_side(model, d):
    out = model(hidden_states=d["packed"], encoder_hidden_states=d["prompt"], pooled_projections=d["pooled"], timestep=d["t"], img_ids=d["img_ids"],
                txt_ids=d["txt_ids"], guidance=d["guidance"] if model.config.guidance_embeds else None, return_dict=False)[0]
    loss = ((out.float() - d["target"].float()) ** 2).mean()
    loss.backward()
    return out.detach(), loss.detach()


def _oracle_side(model, d, params_need_grad, lora=None, lora_scale=1.0):
    P, _, _ = PU.oracle_state(model)
    P = {k: (v.clone().requires_grad_(True) if params_need_grad else v) for k, v in P.items()}
    lp = None if lora is None else {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    f = lambda k: d[k].float()
    out = OF.flux_forward(P, PU.oracle_cfg(model), f("packed"), f("prompt"), f("pooled"), d["t"], d["img_ids"], d["txt_ids"],
                          d["guidance"] if model.config.guidance_embeds else None, lp, lora_scale)
    loss = ((out - f("target")) ** 2).mean()
    loss.backward()
    return out.detach(), loss.detach(), P, lp


@pytest.mark.parametrize("B,lat_h,lat_w,S_txt", [(1, 16, 16, 64), (2, 16, 8, 24)])
def test_lora_path_through_the_emulator_matches_the_oracle(monkeypatch, B, lat_h, lat_w, S_txt):
    """the path the GPU parity tests already pin: if this fails, the emulator (not the engine) is wrong"""
    model = _model(monkeypatch, 2, 2)
    model.add_lora_adapter(rank=16, alpha=16.0, targets="all", init_b_std=0.02)
    d = _inputs(B, lat_h, lat_w, S_txt)
    out, loss = _hip_side(model, d)
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
    print(f"[emu] flux LoRA host sequencing B{B}: pred rel_l2={PU.rel_l2(out, o_out):.3e}, worst adapter gradient rel_l2={worst:.3e}")


def _check_full(model, d):
    out, loss = _hip_side(model, d)
    o_out, o_loss, P, _ = _oracle_side(model, d, True)
    r = PU.rel_l2(out, o_out)
    assert r < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item()), (r, loss.item(), o_loss.item())
    gmax = max(v.grad.norm().item() for v in P.values())
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
    return r, worst, checked


@pytest.mark.parametrize("B,lat_h,lat_w,S_txt,guidance", [(1, 16, 16, 64, True), (2, 16, 8, 24, True), (2, 8, 8, 40, False)])
def test_full_rank_gradients_of_every_parameter_match_the_oracle(monkeypatch, B, lat_h, lat_w, S_txt, guidance):
    model = _model(monkeypatch, 2, 2, guidance=guidance)
    params = model.enable_full_finetune()
    if B == 2 and guidance:
        model._tn_window_bytes = 2 * model.D * 1024            # the modulation matrix's input gradient in 8 row blocks (Flux.1-dev needs 4 at the real 2 GiB window)
    assert len(params) == len(list(model.named_parameters())) and all(p.requires_grad for p in params)
    r, worst, checked = _check_full(model, _inputs(B, lat_h, lat_w, S_txt))
    print(f"[emu] flux full-rank host sequencing B{B} guidance={guidance}: pred rel_l2={r:.3e}; {checked} tensors, worst gradient rel_l2={worst[0]:.3e} at {worst[1]}")
    assert checked > 40


def test_full_rank_checkpointed_gradients_equal_direct_gradients(monkeypatch):
    """a checkpointed segment is re-run from its kept input with the same calls in the same order: every gradient is bit-identical"""
    d = _inputs(2, 16, 8, 24)

    def run(ckpt):
        model = _model(monkeypatch, 2, 3)
        model.enable_full_finetune()
        if ckpt:
            model.enable_gradient_checkpointing()
            model.set_gradient_checkpointing_interval(2)
        _hip_side(model, d)
        return {n: p.grad.clone() for n, p in model.named_parameters()}

    a, b = run(False), run(True)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_full_rank_parameters_are_one_contiguous_run_and_the_fused_optimizer_would_take_one_launch(monkeypatch):
    from simpletuner_amd.training.optimizer import _contiguous_run
    model = _model(monkeypatch, 1, 1)
    params = model.enable_full_finetune()
    assert _contiguous_run([p.data for p in params])
    assert sum(p.numel() for p in params) == model.arena.numel() == model.grad_arena.numel()
    _hip_side(model, _inputs(1, 8, 8, 24))
    assert _contiguous_run([p.grad for p in params])                 # views of ONE private copy of the gradient arena, in arena order


@pytest.mark.parametrize("start,end,ckpt", [(1, 2, False), (1, -2, True), (3, 5, False), (4, -1, False), (0, 1, True)])
def test_tread_routing_through_the_emulator_matches_the_oracle(monkeypatch, start, end, ckpt):
    """the routes of tests/test_flux_model_gpu.py::test_flux_tread_routing_matches_oracle on the CPU: 3 double + 4 single blocks, half of the image tokens routed
    around blocks [start, end] (inside the double stack ending on its last block, across the double / single boundary, starting on single block 0, ending on the
    last block, starting on block 0); LoRA gradients vs the oracle replaying the same permutation; with per-block recomputation the result is bit-identical"""
    from simpletuner_amd.training.tread import ReplayRouter
    d = _inputs(2, 16, 16, 32)
    B, Si = 2, 64
    g = torch.Generator().manual_seed(17)
    perm = torch.stack([torch.randperm(Si, generator=g) for _ in range(B)])
    K = Si - int(round(Si * 0.5))
    rec = {"mask": torch.ones(B, Si, dtype=torch.bool).scatter_(1, perm[:, :K], False), "ids_keep": perm[:, :K], "ids_mask": perm[:, K:], "ids_shuffle": perm,
           "ids_restore": torch.argsort(perm, dim=1)}
    routes = [{"selection_ratio": 0.5, "start_layer_idx": start, "end_layer_idx": end}]

    def run(with_ckpt):
        model = _model(monkeypatch, 3, 4)
        model.add_lora_adapter(rank=8, alpha=8.0, targets="default", init_b_std=0.02)
        model.set_router(ReplayRouter([rec]), routes)
        model.train()
        if with_ckpt:
            model.enable_gradient_checkpointing()
        out, loss = _hip_side(model, d)
        return model, out, loss, {n: p.grad.clone() for n, p in model.named_parameters() if ".lora_" in n}

    model, out, loss, grads = run(False)
    P, lora, scale = PU.oracle_state(model)
    lp = {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    f = lambda k: d[k].float()
    o_out = OF.flux_forward(P, PU.oracle_cfg(model), f("packed"), f("prompt"), f("pooled"), d["t"], d["img_ids"], d["txt_ids"], d["guidance"], lp, scale,
                            tread={"routes": routes, "mask_infos": [rec]})
    o_loss = ((o_out - f("target")) ** 2).mean()
    o_loss.backward()
    assert PU.rel_l2(out, o_out) < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item())
    for name, g_ in grads.items():
        ref = lp[name.split(".lora_")[0]][0 if ".lora_A." in name else 1].grad
        assert PU.rel_l2(g_, ref) < 5e-2, (name, PU.rel_l2(g_, ref))
    if ckpt:
        _, out_c, _, grads_c = run(True)
        assert torch.equal(out, out_c) and all(torch.equal(grads[k], grads_c[k]) for k in grads)


def test_model_type_full_enters_full_rank_training_at_freeze_components(monkeypatch):
    """the reference Trainer never asks for full-rank training explicitly: the loaded model simply has every parameter trainable and `freeze_components()` freezes
    nothing for model_type == "full" (common.py:3700-3709; trainer.py:3668-3673 collects `requires_grad` parameters).  The plugin enters the mode at that call."""
    from types import SimpleNamespace
    EMU.install(monkeypatch)
    from simpletuner_amd.flux import transformer as T
    from simpletuner_amd.flux.model import Flux
    monkeypatch.setattr(T, "_FUSED_QKV", False); monkeypatch.setattr(T, "_BLOCK_ABI", False)
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    plug = Flux(SimpleNamespace(model_type="full", model_family="flux", seed=3), acc)
    plug.model = T.FluxTransformer2DModel(device="cpu", **PU.small_flux_cfg(layers=1, single=1))
    assert not any(p.requires_grad for p in plug.model.parameters())
    plug.freeze_components()
    comp = plug.get_trained_component()
    assert comp.full and all(p.requires_grad for p in comp.parameters())
    assert [id(p) for p in comp.trainable_parameters()] == [id(p) for p in sorted(comp.parameters(), key=lambda q: q.data_ptr())]
    plug.freeze_components()                                   # idempotent
    assert comp.full


def test_attention_masked_training_through_the_emulator_matches_the_oracle(monkeypatch):
    """flux_attention_masked_training (flux/model.py:813-823, flux/transformer.py:170-173, 227-242): the text mask becomes an ADDITIVE +1 on every valid key (the reference
    hands SDPA a float mask); the transformer-level `attention_mask` argument through the per-key bias of the attention kernels, forward and backward"""
    model = _model(monkeypatch, 1, 2)
    model.add_lora_adapter(rank=8, alpha=8.0, targets="default", init_b_std=0.02)
    d = _inputs(2, 16, 16, 32)
    St, Si = 32, 64
    mask = torch.ones(2, St); mask[0, 20:] = 0; mask[1, 9:] = 0
    out = model(hidden_states=d["packed"], encoder_hidden_states=d["prompt"], pooled_projections=d["pooled"], timestep=d["t"], img_ids=d["img_ids"], txt_ids=d["txt_ids"],
                guidance=d["guidance"], attention_mask=mask, return_dict=False)[0]
    loss = ((out.float() - d["target"].float()) ** 2).mean()
    loss.backward()
    P, lora, scale = PU.oracle_state(model)
    lp = {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    kb = torch.ones(2, St + Si); kb[:, :St] = (mask > 0).float()
    f = lambda k: d[k].float()
    o = OF.flux_forward(P, PU.oracle_cfg(model), f("packed"), f("prompt"), f("pooled"), d["t"], d["img_ids"], d["txt_ids"], d["guidance"], lp, scale, key_bias=kb)
    o_plain = OF.flux_forward(P, PU.oracle_cfg(model), f("packed"), f("prompt"), f("pooled"), d["t"], d["img_ids"], d["txt_ids"], d["guidance"],
                              {k: (a.detach(), b.detach()) for k, (a, b) in lp.items()}, scale)
    ((o - f("target")) ** 2).mean().backward()
    assert PU.rel_l2(out, o) < 2e-2 and PU.rel_l2(o_plain, o) > 2e-2          # parity, and the mask matters
    for name, p in model.named_parameters():
        if ".lora_" in name:
            ref = lp[name.split(".lora_")[0]][0 if ".lora_A." in name else 1].grad
            assert PU.rel_l2(p.grad, ref) < 5e-2, name


@pytest.mark.parametrize("masked", [False, True])
def test_fused_projection_path_through_the_emulator_matches_the_oracle(monkeypatch, masked):
    """Flux's DEFAULT host path (head_dim 128, tile-aligned streams: 256 image + 256 text rows per sample, B = 2): RMSNorm + RoPE + the head-major re-layout in the QKV
    GEMM's epilogue (ST355_EPI_QK_NORM_ROPE), the streams as segmented-row problems over the joint buffers, attention from the fused V^T, the backward with the RoPE / RMSNorm
    backward in the dQ / dK epilogues (st355_attn_bwd_rope) recovering x_hat from the roped Q / K — LoRA gradients vs the oracle; and the same step with the fused path
    switched off gives the same answer to bf16 rounding"""
    def run(fused):
        EMU.install(monkeypatch)
        from simpletuner_amd.flux import transformer as T
        monkeypatch.setattr(T, "_FUSED_QKV", fused); monkeypatch.setattr(T, "_BLOCK_ABI", False)
        model = T.FluxTransformer2DModel(device="cpu", **PU.small_flux_cfg(layers=2, single=1))
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if "norm_q" in name or "norm_k" in name or "norm_added" in name:
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                elif name.endswith(".bias"):
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(torch.randn(p.shape, generator=g) / (p.shape[1] ** 0.5))
        model.add_lora_adapter(rank=16, alpha=16.0, targets="default", init_b_std=0.02)
        d = _inputs(2, 32, 32, 256)
        mask = None
        if masked:
            mask = torch.ones(2, 256); mask[0, 160:] = 0; mask[1, 72:] = 0
        calls = []
        orig = EMU.attn_bwd_rope
        monkeypatch.setattr(__import__("simpletuner_amd.ops", fromlist=["ops"]), "attn_bwd_rope", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        out = model(hidden_states=d["packed"], encoder_hidden_states=d["prompt"], pooled_projections=d["pooled"], timestep=d["t"], img_ids=d["img_ids"], txt_ids=d["txt_ids"],
                    guidance=d["guidance"], attention_mask=mask, return_dict=False)[0]
        ((out.float() - d["target"].float()) ** 2).mean().backward()
        return model, d, mask, out.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if ".lora_" in n}, len(calls)

    model, d, mask, out, grads, n_fused = run(True)
    assert n_fused == 3                                        # every block's backward went through the fused attention + RoPE / RMSNorm backward
    _, _, _, out_u, grads_u, n_unfused = run(False)
    assert n_unfused == 0
    P, lora, scale = PU.oracle_state(model)
    lp = {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    kb = None
    if masked:
        kb = torch.ones(2, 512); kb[:, :256] = (mask > 0).float()
    f = lambda k: d[k].float()
    o = OF.flux_forward(P, PU.oracle_cfg(model), f("packed"), f("prompt"), f("pooled"), d["t"], d["img_ids"], d["txt_ids"], d["guidance"], lp, scale, key_bias=kb)
    ((o - f("target")) ** 2).mean().backward()
    assert PU.rel_l2(out, o) < 2e-2 and PU.rel_l2(out_u, o) < 2e-2
    for name, g_ in grads.items():
        ref = lp[name.split(".lora_")[0]][0 if ".lora_A." in name else 1].grad
        assert PU.rel_l2(g_, ref) < 5e-2, (name, PU.rel_l2(g_, ref))
        assert PU.rel_l2(grads_u[name], ref) < 5e-2, name


@pytest.mark.parametrize("layers,single,B,lat_h,lat_w,S_txt", [(2, 2, 1, 16, 16, 64), (1, 2, 2, 32, 32, 256), (0, 2, 1, 16, 8, 24), (2, 2, 2, 16, 8, 24)])      # last: B = 2, 32 / 24 rows per sample
def test_tokenwise_timesteps_through_the_emulator_match_the_oracle(monkeypatch, layers, single, B, lat_h, lat_w, S_txt):
    """TOKENWISE timesteps [B, S_img] (CREPA self-flow; the reference's tests/test_flux_model.py:213-241 hands them to the transformer; oracle branch pinned to the
    executed reference class by tests/test_ref_models_cpu.py): per-token AdaLN rows on the image stream of the double blocks and in norm_out, the token mean on
    the text stream, [mean x S_txt || per token] rows along the single blocks' joint sequence (rows_per_batch = 1); prediction and LoRA gradients (all targets)
    against autograd on the oracle.  B = 2 with 256 image / 256 text rows per sample: segmented problems over per-token gates.  (0, 2): single blocks only."""
    model = _model(monkeypatch, layers, single)
    model.add_lora_adapter(rank=16, alpha=16.0, targets="all", init_b_std=0.02)
    d = _inputs(B, lat_h, lat_w, S_txt)
    Si = (lat_h // 2) * (lat_w // 2)
    d["t"] = torch.rand(B, Si, generator=torch.Generator().manual_seed(8)) * 0.9 + 0.05
    out, loss = _hip_side(model, d)
    _, lora, scale = PU.oracle_state(model)
    o_out, o_loss, _, lp = _oracle_side(model, d, False, lora, scale)
    assert PU.rel_l2(out, o_out) < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item())
    d_flat = dict(d, t=d["t"].mean(dim=1))           # not the batch-wise forward in disguise
    with torch.no_grad():
        out_flat = model(hidden_states=d["packed"], encoder_hidden_states=d["prompt"], pooled_projections=d["pooled"], timestep=d_flat["t"], img_ids=d["img_ids"],
                         txt_ids=d["txt_ids"], guidance=d["guidance"], return_dict=False)[0]
    assert PU.rel_l2(out_flat, o_out) > 5e-2
    worst = 0.0
    for name, p in model.named_parameters():
        if ".lora_" not in name:
            continue
        key, which = name.split(".lora_")
        ref = lp[key][0 if which.startswith("A") else 1].grad
        worst = max(worst, PU.rel_l2(p.grad, ref))
        assert PU.rel_l2(p.grad, ref) < 5e-2, name
    print(f"[emu] flux tokenwise timesteps L{layers}+{single} B{B}: pred rel_l2={PU.rel_l2(out, o_out):.3e}, worst adapter gradient rel_l2={worst:.3e}")


def test_context_target_set_through_the_emulator_matches_the_oracle(monkeypatch):
    """flux_lora_target = "context" (flux/model.py:1263-1271): adapters ONLY on the double blocks' context-stream projections (add_q/k/v_proj, to_add_out); the image
    stream and the single blocks carry no adapter (their backward is the pure data path, block 0 of the double stack still yields its context adapters' gradients)"""
    model = _model(monkeypatch, 2, 2)
    model.add_lora_adapter(rank=16, alpha=16.0, targets="context", init_b_std=0.02)
    names = [n for n, _ in model.named_parameters() if ".lora_" in n]
    assert names and all(("add_" in n or "to_add_out" in n) for n in names) and len(names) == 2 * 2 * 4
    d = _inputs(2, 16, 8, 24)
    out, loss = _hip_side(model, d)
    _, lora, scale = PU.oracle_state(model)
    assert set(lora) == {n.split(".lora_")[0] for n in names}
    o_out, o_loss, _, lp = _oracle_side(model, d, False, lora, scale)
    assert PU.rel_l2(out, o_out) < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item())
    for name, p in model.named_parameters():
        if ".lora_" not in name:
            continue
        key, which = name.split(".lora_")
        ref = lp[key][0 if which.startswith("A") else 1].grad
        assert PU.rel_l2(p.grad, ref) < 5e-2, (name, PU.rel_l2(p.grad, ref))


def test_reference_image_tokens_at_t_zero_through_the_emulator_match_the_oracle(monkeypatch):
    """Kontext (flux/model.py:762-778, 602-618): 16 clean reference-image tokens appended to the 64 scene tokens, their position ids appended to the image ids (ids
    handed over per sample, [B, S, 3]), conditioned on t = 0 through tokenwise timesteps [t ... t | 0 ... 0]; prediction (all 80 tokens) and LoRA gradients of a loss
    on the scene tokens against autograd on the oracle"""
    model = _model(monkeypatch, 1, 2)
    model.add_lora_adapter(rank=16, alpha=16.0, init_b_std=0.02)
    d = _inputs(1, 16, 16, 64)
    g = torch.Generator().manual_seed(21)
    Ss, Sc = d["packed"].shape[1], 16
    cond = torch.randn(1, Sc, 64, generator=g).to(BF16)
    cond_ids = OF.prepare_latent_image_ids(8, 8).clone(); cond_ids[:, 0] = 1.0          # the reference image's own position grid, first id channel 1
    d = dict(d, packed=torch.cat([d["packed"], cond], dim=1), img_ids=torch.cat([d["img_ids"], cond_ids], dim=0)[None],
             t=torch.cat([torch.full((1, Ss), 0.37), torch.zeros(1, Sc)], dim=1), target=torch.cat([d["target"], torch.zeros(1, Sc, 64).to(BF16)], dim=1))

    def loss_of(out):
        return ((out[:, :Ss].float() - d["target"][:, :Ss].float()) ** 2).mean()

    out = model(hidden_states=d["packed"], encoder_hidden_states=d["prompt"], pooled_projections=d["pooled"], timestep=d["t"], img_ids=d["img_ids"],
                txt_ids=d["txt_ids"], guidance=d["guidance"], return_dict=False)[0]
    loss_of(out).backward()
    P, lora, scale = PU.oracle_state(model)
    lp = {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    f = lambda k: d[k].float()
    o_out = OF.flux_forward(P, PU.oracle_cfg(model), f("packed"), f("prompt"), f("pooled"), d["t"], d["img_ids"][0], d["txt_ids"], d["guidance"], lp, scale)
    loss_of(o_out).backward()
    assert out.shape == o_out.shape == (1, Ss + Sc, 64) and PU.rel_l2(out.detach(), o_out.detach()) < 2e-2
    for name, p in model.named_parameters():
        if ".lora_" in name:
            key, which = name.split(".lora_")
            ref = lp[key][0 if which.startswith("A") else 1].grad
            assert PU.rel_l2(p.grad, ref) < 5e-2, (name, PU.rel_l2(p.grad, ref))


def _check_adapter_set(model, d, expect_names):
    names = [n for n, _ in model.named_parameters() if ".lora_" in n]
    keys = {n.split(".lora_")[0] for n in names}
    assert keys == set(expect_names), (sorted(keys ^ set(expect_names)))
    out, loss = _hip_side(model, d)
    _, lora, scale = PU.oracle_state(model)
    o_out, o_loss, _, lp = _oracle_side(model, d, False, lora, scale)
    r = PU.rel_l2(out, o_out)
    assert r < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item()), (r, loss.item(), o_loss.item())
    worst = (0.0, "")
    for name, p in model.named_parameters():
        if ".lora_" not in name:
            continue
        key, which = name.split(".lora_")
        ref = lp[key][0 if which.startswith("A") else 1].grad
        assert p.grad is not None and ref.norm().item() > 0, name
        worst = max(worst, (PU.rel_l2(p.grad, ref), name))
        assert PU.rel_l2(p.grad, ref) < 5e-2, (name, PU.rel_l2(p.grad, ref))
    return r, worst


@pytest.mark.parametrize("which,B,lat_h,lat_w,S_txt,rank", [("all+ffs", 2, 16, 8, 24, 16), ("all+ffs", 1, 16, 16, 64, 16), ("context+ffs", 2, 16, 8, 24, 16), ("all+ffs", 1, 8, 8, 40, 80),
                                                           ("all+ffs+embedder", 2, 16, 8, 24, 16), ("ai-toolkit", 2, 16, 8, 24, 16), ("ai-toolkit", 1, 16, 16, 64, 16)])
def test_feed_forward_target_sets_through_the_emulator_match_the_oracle(monkeypatch, which, B, lat_h, lat_w, S_txt, rank):
    """flux_lora_target = "all+ffs" / "context+ffs" (flux/model.py:1272-1301): adapters on ff.net.0.proj / ff.net.2 / ff_context.net.* of the double blocks and on
    proj_mlp / proj_out of the single blocks, next to the attention projections of the set.  proj_out reads [attn | mlp] as two K segments: its adapter's T = x A^T and
    dA = U^T x walk the same two segments; its low-rank term leaves as a gated-residual launch of its own.  Rank 80 walks the rank-space kernels in 64-column slabs."""
    model = _model(monkeypatch, 2, 2)
    model.add_lora_adapter(rank=rank, alpha=float(rank), targets=which, init_b_std=0.02)
    r, worst = _check_adapter_set(model, _inputs(B, lat_h, lat_w, S_txt), OF.lora_targets(PU.oracle_cfg(model), which))
    print(f"[emu] flux {which} rank {rank} B{B}: pred rel_l2={r:.3e}, worst adapter gradient {worst[1]} rel_l2={worst[0]:.3e}")


@pytest.mark.parametrize("which,single", [("nano", 9), ("tiny", 22)])
def test_single_layer_target_sets_through_the_emulator_match_the_oracle(monkeypatch, which, single):
    """flux_lora_target = "nano" / "tiny" (flux/model.py:1363-1375): single_transformer_blocks.7(.20).proj_out and nothing else.  The backward stops below single
    block 7: no block upstream of it runs a backward launch (counted on the emulator's call log)."""
    model = _model(monkeypatch, 1, single)
    model.add_lora_adapter(rank=16, alpha=16.0, targets=which, init_b_std=0.02)
    assert model._bwd_stop == 1 + 7
    seen = []
    from simpletuner_amd.flux import transformer as T
    real_single, real_double = T.FluxTransformer2DModel._single_bwd, T.FluxTransformer2DModel._double_bwd
    monkeypatch.setattr(T.FluxTransformer2DModel, "_single_bwd", lambda self, li, *a, **k: (seen.append(("s", li)), real_single(self, li, *a, **k))[1])
    monkeypatch.setattr(T.FluxTransformer2DModel, "_double_bwd", lambda self, li, *a, **k: (seen.append(("d", li)), real_double(self, li, *a, **k))[1])
    r, worst = _check_adapter_set(model, _inputs(2, 8, 8, 24), OF.lora_targets(PU.oracle_cfg(model), which))
    assert seen == [("s", li) for li in range(single - 1, 6, -1)], seen
    print(f"[emu] flux {which}: pred rel_l2={r:.3e}, worst adapter gradient {worst[1]} rel_l2={worst[0]:.3e}; backward ran single blocks {single - 1}..7 only")


def test_single_layer_target_sets_refuse_a_model_without_that_block(monkeypatch):
    model = _model(monkeypatch, 1, 4)
    with pytest.raises(ValueError, match="single_transformer_blocks.7.proj_out"):
        model.add_lora_adapter(rank=16, targets="nano")


def test_feed_forward_adapters_under_tread_routing_and_recomputation(monkeypatch):
    """all+ffs with half of the image tokens routed around double block 1 .. single block 1 (across the stack boundary): adapter gradients vs the oracle replaying
    the permutation; with per-block recomputation the forward and every gradient are bit-identical to the run that kept its activations"""
    from simpletuner_amd.training.tread import ReplayRouter
    d = _inputs(2, 16, 16, 32)
    B, Si = 2, 64
    perm = torch.stack([torch.randperm(Si, generator=torch.Generator().manual_seed(17 + b)) for b in range(B)])
    K = Si - int(round(Si * 0.5))
    rec = {"mask": torch.ones(B, Si, dtype=torch.bool).scatter_(1, perm[:, :K], False), "ids_keep": perm[:, :K], "ids_mask": perm[:, K:], "ids_shuffle": perm,
           "ids_restore": torch.argsort(perm, dim=1)}
    routes = [{"selection_ratio": 0.5, "start_layer_idx": 1, "end_layer_idx": 3}]

    def run(with_ckpt):
        model = _model(monkeypatch, 2, 3)
        model.add_lora_adapter(rank=8, alpha=8.0, targets="all+ffs", init_b_std=0.02)
        model.set_router(ReplayRouter([rec]), routes)
        model.train()
        if with_ckpt:
            model.enable_gradient_checkpointing()
        out, loss = _hip_side(model, d)
        return model, out, loss, {n: p.grad.clone() for n, p in model.named_parameters() if ".lora_" in n}

    model, out, loss, grads = run(False)
    P, lora, scale = PU.oracle_state(model)
    lp = {k: (a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for k, (a, b) in lora.items()}
    f = lambda k: d[k].float()
    o_out = OF.flux_forward(P, PU.oracle_cfg(model), f("packed"), f("prompt"), f("pooled"), d["t"], d["img_ids"], d["txt_ids"], d["guidance"], lp, scale,
                            tread={"routes": routes, "mask_infos": [rec]})
    ((o_out - f("target")) ** 2).mean().backward()
    assert PU.rel_l2(out, o_out) < 2e-2
    assert any("ff.net.2" in n for n in grads) and any("proj_out" in n for n in grads)
    for name, g_ in grads.items():
        ref = lp[name.split(".lora_")[0]][0 if ".lora_A." in name else 1].grad
        assert PU.rel_l2(g_, ref) < 5e-2, (name, PU.rel_l2(g_, ref))
    _, out_c, _, grads_c = run(True)
    assert torch.equal(out, out_c) and all(torch.equal(grads[k], grads_c[k]) for k in grads)


def test_feed_forward_adapters_with_tokenwise_timesteps(monkeypatch):
    """all+ffs under per-token timesteps: the feed-forward gates are per-token rows on the image stream and along the single blocks' joint sequence — proj_out's
    low-rank launch takes the same per-token gate rows as the projection itself"""
    model = _model(monkeypatch, 1, 2)
    model.add_lora_adapter(rank=16, alpha=16.0, targets="all+ffs", init_b_std=0.02)
    d = _inputs(1, 16, 8, 24)
    d["t"] = torch.rand(1, 32, generator=torch.Generator().manual_seed(8)) * 0.9 + 0.05
    r, worst = _check_adapter_set(model, d, OF.lora_targets(PU.oracle_cfg(model), "all+ffs"))
    print(f"[emu] flux all+ffs tokenwise: pred rel_l2={r:.3e}, worst adapter gradient {worst[1]} rel_l2={worst[0]:.3e}")


def test_nano_with_recomputation_is_bit_identical_and_recomputes_only_what_it_differentiates(monkeypatch):
    """'nano' + gradient checkpointing: the segments below single block 7 are never re-run (the backward returns there)"""
    d = _inputs(1, 8, 8, 24)
    from simpletuner_amd.flux import transformer as T

    def run(with_ckpt):
        model = _model(monkeypatch, 1, 9)
        model.add_lora_adapter(rank=16, alpha=16.0, targets="nano", init_b_std=0.02)
        model.train()
        fwd_calls, dbl_saves = [], []
        if with_ckpt:
            model.enable_gradient_checkpointing()
        real, real_d = T.FluxTransformer2DModel._single_fwd, T.FluxTransformer2DModel._double_fwd
        monkeypatch.setattr(T.FluxTransformer2DModel, "_single_fwd", lambda self, bi, x, env, save: (fwd_calls.append((bi, save)), real(self, bi, x, env, save))[1])
        monkeypatch.setattr(T.FluxTransformer2DModel, "_double_fwd", lambda self, bi, i_, t_, env, save: (dbl_saves.append(save), real_d(self, bi, i_, t_, env, save))[1])
        out, _ = _hip_side(model, d)
        monkeypatch.setattr(T.FluxTransformer2DModel, "_single_fwd", real)
        monkeypatch.setattr(T.FluxTransformer2DModel, "_double_fwd", real_d)
        return out, {n: p.grad.clone() for n, p in model.named_parameters() if ".lora_" in n}, fwd_calls, dbl_saves

    out, grads, calls_direct, dbl_direct = run(False)
    # without recomputation: only single blocks 7 and 8 keep their activations; the double block and single blocks 0 .. 6 run as in inference
    assert calls_direct == [(bi, bi >= 7) for bi in range(9)] and dbl_direct == [False], (calls_direct, dbl_direct)
    out_c, grads_c, calls, _ = run(True)
    assert torch.equal(out, out_c) and all(torch.equal(grads[k], grads_c[k]) for k in grads)
    assert all(not save for (_, save) in calls[:9])
    recomputed = [bi for (bi, save) in calls[9:]]            # the first 9 calls are the forward itself
    assert recomputed == [8, 7], calls                        # per-block segments: blocks 6 .. 0 are never re-run


@pytest.mark.parametrize("which", ["all", "context", "all+ffs", "context+ffs", "all+ffs+embedder", "ai-toolkit", "nano", "tiny"])
def test_adapter_names_are_the_modules_peft_wraps_in_the_executed_reference(monkeypatch, which):
    """the product's adapter parameters (and so the keys of the saved LoRA file) against tests/golden/ref_flux_lora_sets.pt: the module names peft's target_modules
    rule selected on the reference's FluxTransformer2DModel from the reference's own `flux_lora_target` list, for a model of the same depth"""
    import os
    G = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ref_flux_lora_sets.pt"), weights_only=False)[which]
    model = _model(monkeypatch, G["config"]["num_layers"], G["config"]["num_single_layers"])
    model.add_lora_adapter(rank=4, alpha=8.0, targets=which)
    names = [n for n, _ in model.named_parameters() if ".lora_" in n]
    assert {n.split(".lora_")[0] for n in names} == set(G["lora_targets"])
    assert len(names) == 2 * len(G["lora_targets"]) and all(n.endswith((".lora_A.default.weight", ".lora_B.default.weight")) for n in names)
    shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
    for t in G["lora_targets"]:          # peft shapes: A [r, in_features], B [out_features, r]
        w = model.get_parameter(t + ".weight") if t + ".weight" in shapes else None
        a, b = shapes[t + ".lora_A.default.weight"], shapes[t + ".lora_B.default.weight"]
        assert a[0] == 4 and b[1] == 4
        if w is not None:
            assert (b[0], a[1]) == tuple(w.shape), (t, a, b, tuple(w.shape))


def test_embedder_adapter_on_a_model_without_double_blocks(monkeypatch):
    """all+ffs+embedder with single blocks only: x_embedder's dy is the image rows of the joint gradient at single block 0's input"""
    model = _model(monkeypatch, 0, 2)
    model.add_lora_adapter(rank=16, alpha=16.0, targets="all+ffs+embedder", init_b_std=0.02)
    r, worst = _check_adapter_set(model, _inputs(2, 16, 8, 24), OF.lora_targets(PU.oracle_cfg(model), "all+ffs+embedder"))
    print(f"[emu] flux all+ffs+embedder, single blocks only: pred rel_l2={r:.3e}, worst adapter gradient {worst[1]} rel_l2={worst[0]:.3e}")


@pytest.mark.parametrize("fmt", ["diffusers", "comfyui"])
def test_lora_file_round_trip_with_the_widest_adapter_set(monkeypatch, tmp_path, fmt):
    """save_lora_weights / load_lora_weights through the Flux plugin with flux_lora_target = "all+ffs+embedder": every wrapped module lands in the file under its peft name
    (`transformer.<module>.lora_A.weight` / `.lora_B.weight`; "comfyui" adds one `.alpha` per module) and comes back into a freshly initialised adapter bit for bit"""
    from types import SimpleNamespace

    from safetensors.torch import load_file

    from simpletuner_amd.flux.model import Flux
    model = _model(monkeypatch, 1, 2)
    model.add_lora_adapter(rank=4, alpha=8.0, targets="all+ffs+embedder", init_b_std=0.05)
    plug = Flux(SimpleNamespace(lora_format=fmt, lora_alpha=8.0, lora_rank=4, flux_lora_target="all+ffs+embedder", model_type="lora"), SimpleNamespace(device=torch.device("cpu")))
    plug.model = model
    path = plug.save_lora_weights(str(tmp_path))
    flat = load_file(path)
    want_modules = set(OF.lora_targets(PU.oracle_cfg(model), "all+ffs+embedder"))
    got_modules = {k[len("transformer."):].rsplit(".lora_", 1)[0] for k in flat if ".lora_" in k}
    assert got_modules == want_modules and all(k.startswith("transformer.") for k in flat)
    assert len([k for k in flat if k.endswith(".alpha")]) == (len(want_modules) if fmt == "comfyui" else 0)
    before = {n: p.detach().clone() for n, p in model.named_parameters() if ".lora_" in n}
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".lora_" in n:
                p.zero_()
    plug.load_lora_weights(input_dir=str(tmp_path))
    for n, p in model.named_parameters():
        if ".lora_" in n:
            assert torch.equal(p, before[n]), n


def test_modulation_adapters_recompute_bit_identically_and_refuse_what_is_not_built(monkeypatch):
    """flux_lora_target = "ai-toolkit" (flux/model.py:1340-1362: all+ffs + norm1.linear / norm1_context.linear / norm.linear): the modulation Linears' adapters are ONE group
    over the fused modulation GEMM; their dy is the per-sample modulation-row gradient (shift / scale / gate column sums the frozen-base backward otherwise never forms).
    With per-block recomputation every gradient is bit-identical; tokenwise timesteps and TREAD routes are refused for this set."""
    d = _inputs(2, 16, 8, 24)

    def run(ckpt):
        model = _model(monkeypatch, 2, 2)
        model.add_lora_adapter(rank=8, alpha=8.0, targets="ai-toolkit", init_b_std=0.02)
        model.train()
        if ckpt:
            model.enable_gradient_checkpointing()
        out, _ = _hip_side(model, d)
        return model, out, {n: p.grad.clone() for n, p in model.named_parameters() if ".lora_" in n}

    model, out, grads = run(False)
    mods = [n for n in grads if ".norm1.linear." in n or ".norm1_context.linear." in n or ".norm.linear." in n]
    assert len(mods) == 2 * (2 * 2 + 2) and all(grads[n].abs().max() > 0 for n in mods)
    assert model.mod_lora is not None and len(model.mod_lora.targets) == 6 and model._dmod is None
    _, out_c, grads_c = run(True)
    assert torch.equal(out, out_c) and all(torch.equal(grads[k], grads_c[k]) for k in grads)
    tok = dict(d, t=torch.rand(2, 32) * 0.8 + 0.1)
    with pytest.raises(NotImplementedError, match="ai-toolkit"):
        _hip_side(model, tok)
    from simpletuner_amd.training.tread import ReplayRouter
    perm = torch.stack([torch.randperm(32, generator=torch.Generator().manual_seed(b)) for b in range(2)])
    rec = {"mask": torch.ones(2, 32, dtype=torch.bool).scatter_(1, perm[:, :16], False), "ids_keep": perm[:, :16], "ids_mask": perm[:, 16:], "ids_shuffle": perm,
           "ids_restore": torch.argsort(perm, dim=1)}
    model.set_router(ReplayRouter([rec]), [{"selection_ratio": 0.5, "start_layer_idx": 1, "end_layer_idx": 2}])
    with pytest.raises(NotImplementedError, match="ai-toolkit"):
        _hip_side(model, d)


@pytest.mark.parametrize("which,layers,single,B,lat_h,lat_w,S_txt,ckpt", [
    ("all+ffs", 2, 0, 2, 8, 8, 40, True), ("context+ffs", 2, 0, 2, 32, 32, 256, False), ("ai-toolkit", 2, 0, 1, 16, 8, 24, True), ("all+ffs+embedder", 2, 0, 2, 32, 32, 256, True),
    ("all+ffs", 0, 2, 2, 32, 32, 256, False), ("ai-toolkit", 0, 2, 2, 8, 8, 40, True), ("ai-toolkit", 2, 3, 2, 32, 32, 256, True), ("all+ffs+embedder", 2, 3, 1, 16, 8, 24, True),
    ("context", 2, 0, 1, 16, 8, 24, False), ("all", 2, 0, 2, 8, 8, 40, True)])
def test_adapter_sets_across_depths_shapes_and_recomputation(monkeypatch, which, layers, single, B, lat_h, lat_w, S_txt, ckpt):
    """a sample of the (set x depth x shape x recomputation) grid swept while the sets were built (70 configurations, none failed): double-only and single-only models,
    tile-aligned and ragged streams, per-block recomputation.  In a model WITHOUT single blocks the context stream of the last double block is discarded, so adapters that
    feed only that output (its add_q_proj, to_add_out, ff_context.*) have an exactly zero gradient in the reference — and must have one here."""
    model = _model(monkeypatch, layers, single)
    model.add_lora_adapter(rank=8, alpha=8.0, targets=which, init_b_std=0.02)
    model.train()
    if ckpt:
        model.enable_gradient_checkpointing()
    d = _inputs(B, lat_h, lat_w, S_txt)
    out, loss = _hip_side(model, d)
    _, lora, scale = PU.oracle_state(model)
    assert set(lora) == set(OF.lora_targets(PU.oracle_cfg(model), which))
    o_out, o_loss, _, lp = _oracle_side(model, d, False, lora, scale)
    assert PU.rel_l2(out, o_out) < 2e-2 and abs(loss.item() - o_loss.item()) < 2e-3 * max(1.0, o_loss.item())
    zeros = 0
    for name, p in model.named_parameters():
        if ".lora_" not in name:
            continue
        key, ab = name.split(".lora_")
        ref = lp[key][0 if ab.startswith("A") else 1].grad
        if ref.norm().item() == 0:
            zeros += 1
            assert p.grad is None or p.grad.abs().max().item() == 0, name
        else:
            assert PU.rel_l2(p.grad, ref) < 5e-2, (name, PU.rel_l2(p.grad, ref))
    assert (zeros > 0) == (single == 0 and which != "default" and any(t in which for t in ("all", "context", "ai-toolkit")))


class _Reached(Exception):
    pass


@pytest.mark.parametrize("entry", ["block_flux_double_fwd", "block_flux_single_fwd", "block_flux_double_bwd", "block_flux_single_bwd"])
def test_block_entry_call_sites_build_their_arguments_on_the_default_path_and_step_aside_for_wider_sets(monkeypatch, entry):
    """The Flux block-level C entry points cannot run on the CPU, but their CALL SITES can be driven up to the call: with the entry point replaced by a stub that records
    its keyword arguments, the default adapter set on tile-aligned streams reaches it (every argument expression of the call site is evaluated — a name that does not exist
    would fail here, not on the GPU box), and the sets the entry points do not know (feed-forward / modulation adapters) never reach it."""
    from simpletuner_amd import ops as real_ops

    def build(which, abi_fwd, abi_bwd):
        EMU.install(monkeypatch)
        from simpletuner_amd.flux import transformer as T
        monkeypatch.setattr(T, "_FUSED_QKV", True)
        model = T.FluxTransformer2DModel(device="cpu", **PU.small_flux_cfg(layers=2, single=2))
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if "norm_q" in name or "norm_k" in name or "norm_added" in name:
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                elif name.endswith(".bias"):
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(torch.randn(p.shape, generator=g) / (p.shape[1] ** 0.5))
        model.add_lora_adapter(rank=16, alpha=16.0, targets=which, init_b_std=0.02)
        return T, model

    seen = {}

    def stub(name):
        def f(*a, **k):
            seen[name] = dict(k)
            raise _Reached(name)
        return f

    fwd = entry.endswith("_fwd")
    d = _inputs(2, 32, 32, 256)
    for which, expect in (("default", True), ("all+ffs", False), ("ai-toolkit", False)):
        seen.clear()
        T, model = build(which, fwd, not fwd)
        monkeypatch.setattr(T, "_BLOCK_ABI", fwd)                    # forward entries: on from the start; backward entries: the forward runs host-sequenced (fused projection)
        for n in ("block_flux_double_fwd", "block_flux_single_fwd", "block_flux_double_bwd", "block_flux_single_bwd"):
            monkeypatch.setattr(real_ops, n, stub(n))
        only = "double" if "double" in entry else "single"
        monkeypatch.setattr(T, "_BLOCK_ABI_ONLY", only)
        try:
            out = model(hidden_states=d["packed"], encoder_hidden_states=d["prompt"], pooled_projections=d["pooled"], timestep=d["t"], img_ids=d["img_ids"],
                        txt_ids=d["txt_ids"], guidance=d["guidance"], return_dict=False)[0]
            if not fwd:
                monkeypatch.setattr(T, "_BLOCK_ABI", True)
                ((out.float() - d["target"].float()) ** 2).mean().backward()
            reached = False
        except _Reached as e:
            reached = str(e) == entry
        assert reached == expect, (entry, which, sorted(seen))
        if expect:
            kw = seen[entry]
            assert kw["B"] == 2 and kw["D"] == model.D and all(v is None or isinstance(v, (int, float, torch.Tensor, list)) for v in kw.values()), sorted(kw)


def test_rope_tables_follow_the_id_contents_not_the_tensor_address(monkeypatch):
    """Two Kontext-shaped calls with the same token counts and different reference-image ids (a transposed position grid: what a 832x1216 reference image after a
    1216x832 one looks like).  The tables must follow the contents: without a layout key they are computed from the ids every call (nothing cached), with a key the
    cache is hit only for the layout the key names, and it stays bounded."""
    model = _model(monkeypatch, 1, 1)
    d = _inputs(1, 16, 16, 64)
    a = OF.prepare_latent_image_ids(4, 16).clone(); a[:, 0] = 1.0
    b = OF.prepare_latent_image_ids(16, 4).clone(); b[:, 0] = 1.0                     # same 16 tokens, transposed grid
    ids_a, ids_b = torch.cat([d["img_ids"], a], 0), torch.cat([d["img_ids"], b], 0)
    assert ids_a.shape == ids_b.shape and not torch.equal(ids_a, ids_b)

    def oracle_tables(ids):
        full = torch.cat([d["txt_ids"], ids], 0).double()
        cs = []
        for i, dim in enumerate(model.config.axes_dims_rope):
            f = torch.outer(full[:, i], 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim)))
            cs.append(f.cos().repeat_interleave(2, dim=1).float())
        return torch.cat(cs, -1)

    model._rope_layout_key = None
    buf = ids_a.clone()
    ca = model._rope(d["txt_ids"], buf)[0].clone()
    buf.copy_(ids_b)                                                                  # same storage, same shape, new contents: the allocator-reuse case
    cb = model._rope(d["txt_ids"], buf)[0].clone()
    assert torch.equal(ca, oracle_tables(ids_a)) and torch.equal(cb, oracle_tables(ids_b)) and not torch.equal(ca, cb)
    assert len(model._rope_cache) == 0
    model._rope_layout_key = ("layout", "a")
    assert torch.equal(model._rope(d["txt_ids"], ids_a)[0], ca) and len(model._rope_cache) == 1
    assert model._rope(d["txt_ids"], ids_a)[0] is model._rope(d["txt_ids"], ids_a)[0]          # hit
    model._rope_layout_key = ("layout", "b")
    assert torch.equal(model._rope(d["txt_ids"], ids_b)[0], cb)
    for i in range(3 * model._ROPE_CACHE_MAX):
        model._rope_layout_key = ("layout", i)
        model._rope(d["txt_ids"], ids_a)
    assert len(model._rope_cache) <= model._ROPE_CACHE_MAX
    model._rope_layout_key = None


def test_flux_plugin_names_the_rope_layout_by_content():
    """the plugin's layout key: latent grid + text length, plus a digest of the reference-image ids when they arrive on the host (never tensor identity)"""
    from types import SimpleNamespace

    import simpletuner_amd.flux.model as FM
    m = FM.Flux.__new__(FM.Flux)
    m.config, m.accelerator, m._ids_cache = SimpleNamespace(model_flavour="kontext"), SimpleNamespace(device=torch.device("cpu")), {}
    keys = []
    m.model = lambda **kw: (keys.append(kw["rope_layout_key"]), (torch.zeros(1, kw["hidden_states"].shape[1], 64, dtype=torch.bfloat16),))[1]
    m.get_trained_component = lambda: SimpleNamespace(config=SimpleNamespace(guidance_embeds=False))
    orig, orig_pack = FM._UnpackFn.apply, FM.pack_latents
    FM._UnpackFn.apply = staticmethod(lambda packed, h, w: torch.zeros(packed.shape[0], 16, h // 8, w // 8))
    FM.pack_latents = lambda x: x.reshape(x.shape[0], -1, 64)
    try:
        batch = {"noisy_latents": torch.randn(1, 16, 4, 4), "latents": torch.randn(1, 16, 4, 4), "timesteps": torch.tensor([500.0]),
                 "prompt_embeds": torch.randn(1, 3, 16), "add_text_embeds": torch.randn(1, 8)}
        m._model_predict_single(dict(batch))
        ia, ib = torch.zeros(1, 2, 3), torch.zeros(1, 2, 3); ib[0, 1, 2] = 1.0
        for ids in (ia, ib, ia.clone()):
            m._model_predict_single(dict(batch, conditioning_packed_latents=torch.randn(1, 2, 64), conditioning_ids=ids))
    finally:
        FM._UnpackFn.apply, FM.pack_latents = orig, orig_pack
    assert keys[0] == ("flux_ids", 4, 4, 3)
    assert keys[1] != keys[2] and keys[1] == keys[3] and keys[1][:4] == keys[0] and all(k is not None for k in keys)
