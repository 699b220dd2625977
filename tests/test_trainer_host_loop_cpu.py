"""The whole training step's HOST LOGIC on the CPU (kernels replaced by tests/ops_emulator.py): `Trainer.train_step` — prepare_batch (noising), model_predict (the Flux
engine, pack / unpack), loss_with_logs (fused loss + gradient), backward, gradient clipping, the fused optimizer over the flat arena, LR schedule, EMA — against the oracle
stepped by `torch.optim.AdamW` (the GPU form of this run is tests/test_flux_model_gpu.py::test_flux_loss_curve_matches_oracle_adamw), and a full-rank run with EMA + clipping."""
from types import SimpleNamespace

import torch

from tests import ops_emulator as EMU
from tests import parity_utils as PU


def _acc():
    return SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True, gradient_accumulation_steps=1, sync_gradients=True,
                           backward=lambda loss: loss.backward(), wait_for_everyone=lambda: None)


def _build(monkeypatch, layers, single, B, lat_h, lat_w, S_txt, **cfg_kw):
    EMU.install(monkeypatch)
    from simpletuner_amd.flux import transformer as T
    from simpletuner_amd.flux.model import Flux
    from simpletuner_amd.training.trainer import Trainer, default_config
    monkeypatch.setattr(T, "_FUSED_QKV", False); monkeypatch.setattr(T, "_BLOCK_ABI", False)
    cfg = default_config(train_batch_size=B, seed=3, flow_schedule_shift=3.0, **cfg_kw)
    acc = _acc()
    plugin = Flux(cfg, acc)
    plugin.load_model(**PU.small_flux_cfg(layers=layers, single=single))
    if cfg.model_type == "lora":
        plugin.add_lora_adapter()
    else:
        plugin.freeze_components()                       # model_type == "full": the mode is entered here, as under the reference Trainer
    trainer = Trainer(cfg, plugin, acc)
    cpu, devt = PU.make_inputs(B, lat_h, lat_w, S_txt, 128, 64, "cpu", seed=3)
    sig = devt["sigmas"]
    plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
    return plugin, trainer, cpu, devt


def _batch(devt):
    return {"latent_batch": devt["latents"], "prompt_embeds": devt["prompt"], "add_text_embeds": devt["pooled"], "noise": devt["noise"]}


def test_lora_loss_curve_of_the_whole_step_matches_the_oracle_under_adamw(monkeypatch):
    plugin, trainer, cpu, devt = _build(monkeypatch, 1, 1, 2, 16, 16, 32, lora_rank=8, lora_init_b_std=0.02, learning_rate=2e-3)
    model = plugin.get_trained_component()
    P, lora, scale = PU.oracle_state(model)
    ocfg = PU.oracle_cfg(model)
    names = sorted(lora)
    params = {k: (torch.nn.Parameter(lora[k][0].clone()), torch.nn.Parameter(lora[k][1].clone())) for k in names}
    opt = torch.optim.AdamW([t for k in names for t in params[k]], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    hip, ora = [], []
    for _ in range(6):
        hip.append(trainer.train_step(_batch(devt)).item())
        opt.zero_grad()
        s = cpu["sigmas"].view(-1, 1, 1, 1)
        noisy = ((1 - s) * cpu["latents"] + s * cpu["noise"]).to(torch.bfloat16).float()
        target = (cpu["noise"] - cpu["latents"]).to(torch.bfloat16).float()
        pred = PU.OF.flux_model_predict(P, ocfg, noisy, cpu["prompt"], cpu["pooled"], cpu["sigmas"] * 1000.0, 1.0, lora={k: params[k] for k in names}, lora_scale=scale)
        l = ((pred - target) ** 2).mean(dim=(1, 2, 3)).mean()
        l.backward(); opt.step()
        ora.append(l.item())
    d = max(abs(a - b) for a, b in zip(hip, ora))
    print(f"[emu] whole-step loss curve (6 AdamW steps): max |delta| {d:.2e}; emulated {[round(x, 4) for x in hip]}")
    assert d < 2e-3 * max(1.0, max(ora)) and hip[-1] < hip[0]
    assert trainer.state["global_step"] == 6
    assert all(p.grad is None for p in model.trainable_parameters())          # zero_grad(set_to_none=True) after the step (trainer.py:7253)


def test_full_rank_steps_with_ema_and_norm_clipping(monkeypatch):
    plugin, trainer, cpu, devt = _build(monkeypatch, 1, 1, 2, 8, 8, 24, model_type="full", learning_rate=2e-4, use_ema=True, ema_decay=0.9, max_grad_norm=0.5)
    model = plugin.get_trained_component()
    assert model.full and trainer.ema_model is not None
    before = model.arena.clone()
    losses = [trainer.train_step(_batch(devt)).item() for _ in range(4)]
    print(f"[emu] full-rank whole-step losses (EMA + clip): {[round(x, 4) for x in losses]}")
    assert losses[-1] < losses[0]
    assert not torch.equal(before, model.arena)                               # ONE fused optimizer step over the arena moved the weights
    assert float(trainer.last_grad_norm) > 0.5                                # the un-clipped norm is logged; the clip coefficient was applied on the arena
    # the shadow lags the weights: s_k = s_{k-1} - (1 - d)(s_{k-1} - p_k)
    sh = trainer.ema_model.shadow_flat if hasattr(trainer.ema_model, "shadow_flat") else None
    if sh is not None:
        lag = (sh.float() - model.arena.float()).abs().max().item()
        moved = (before.float() - model.arena.float()).abs().max().item()
        assert 0.0 < lag < moved


def test_full_rank_checkpoint_resume_continues_bit_for_bit(monkeypatch, tmp_path):
    """save_state after two steps (the trained component as `<dir>/transformer/diffusion_pytorch_model.safetensors` + config.json — what `save_pretrained` writes —, optimizer
    moments, step counters), load_state into a freshly built trainer with different initial weights, two more steps: the same losses and the same arena as the
    uninterrupted run"""
    import json
    import os
    plugin, trainer, cpu, devt = _build(monkeypatch, 1, 1, 2, 8, 8, 24, model_type="full", learning_rate=2e-4)
    for _ in range(2):
        trainer.train_step(_batch(devt))
    ck = str(tmp_path / "checkpoint-2")
    trainer.save_state(ck)
    tail_a = [trainer.train_step(_batch(devt)).item() for _ in range(2)]
    arena_a = plugin.get_trained_component().arena.clone()
    sub = os.path.join(ck, "transformer")
    assert os.path.exists(os.path.join(sub, "diffusion_pytorch_model.safetensors"))
    assert json.load(open(os.path.join(sub, "config.json")))["_class_name"] == "FluxTransformer2DModel"
    from safetensors.torch import load_file
    sd = load_file(os.path.join(sub, "diffusion_pytorch_model.safetensors"))
    assert set(sd) == {n for n, _ in plugin.get_trained_component().named_parameters()}          # diffusers checkpoint keys, one tensor each

    plugin2, trainer2, _, _ = _build(monkeypatch, 1, 1, 2, 8, 8, 24, model_type="full", learning_rate=2e-4)
    with torch.no_grad():
        plugin2.get_trained_component().arena.add_(0.25)                                         # a different start: everything must come from the checkpoint
    trainer2.load_state(ck)
    assert trainer2.state["global_step"] == 2
    tail_b = [trainer2.train_step(_batch(devt)).item() for _ in range(2)]
    assert tail_a == tail_b
    assert torch.equal(arena_a, plugin2.get_trained_component().arena)


def test_sd3_full_fine_tune_checkpoint_resume_continues_bit_for_bit(monkeypatch, tmp_path):
    """the same round trip for the SD3 full fine-tune (BASELINE.json configs[3]): the saved component carries the position table next to the parameters"""
    import os
    EMU.install(monkeypatch)
    from simpletuner_amd.sd3.model import SD3
    from simpletuner_amd.training.trainer import Trainer, default_config
    arch = dict(sample_size=32, num_layers=2, num_attention_heads=2, attention_head_dim=64, joint_attention_dim=128, caption_projection_dim=128, pooled_projection_dim=64,
                pos_embed_max_size=24)

    def build():
        cfg = default_config(model_family="sd3", model_type="full", train_batch_size=2, seed=5, learning_rate=2e-4, flow_schedule_shift=3.0)
        plugin = SD3(cfg, _acc())
        plugin.load_model(**arch)
        plugin.freeze_components()
        trainer = Trainer(cfg, plugin, plugin.accelerator)
        _, devt = PU.make_inputs(2, 16, 16, 24, 128, 64, "cpu", seed=5)
        sig = devt["sigmas"]
        plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
        return plugin, trainer, devt

    plugin, trainer, devt = build()
    assert plugin.get_trained_component().full
    for _ in range(2):
        trainer.train_step(_batch(devt))
    ck = str(tmp_path / "checkpoint-2")
    trainer.save_state(ck)
    tail_a = [trainer.train_step(_batch(devt)).item() for _ in range(2)]
    from safetensors.torch import load_file
    sd = load_file(os.path.join(ck, "transformer", "diffusion_pytorch_model.safetensors"))
    assert "pos_embed.pos_embed" in sd and "pos_embed.proj.weight" in sd and sd["pos_embed.proj.weight"].dim() == 4       # the Conv2d tensor keeps its checkpoint shape
    plugin2, trainer2, _ = build()
    with torch.no_grad():
        plugin2.get_trained_component().arena.add_(0.25)
    trainer2.load_state(ck)
    tail_b = [trainer2.train_step(_batch(devt)).item() for _ in range(2)]
    assert tail_a == tail_b and torch.equal(plugin.get_trained_component().arena, plugin2.get_trained_component().arena)


def test_lora_under_adamw_bf16_trains_bf16_adapter_values(monkeypatch):
    """the reference's LoRA examples train with optimizer=adamw_bf16 over bf16 adapter weights (simpletuner/examples/sd3.peft-lora/config.json;
    optimizers/adamw_bfloat16/__init__.py:66 asserts bf16): over the engine's fp32 adapter arena the trainer keeps a bf16 arena the optimizer steps and mirrors it
    into the fp32 one — every trained value stays a bf16 number, the gradients reach the optimizer rounded to bf16, the loss falls"""
    plugin, trainer, cpu, devt = _build(monkeypatch, 1, 1, 2, 16, 16, 32, lora_rank=8, lora_init_b_std=0.02, learning_rate=2e-3, optimizer="adamw_bf16")
    from simpletuner_amd.training.optimizer import St355AdamWBF16
    model = plugin.get_trained_component()
    sh = trainer._bf16_shadow
    assert isinstance(trainer.optimizer, St355AdamWBF16) and sh is not None
    assert all(q.dtype == torch.bfloat16 and q.shape == p.shape for q, p in zip(sh.params, trainer.params))
    assert torch.equal(model.lora_flat[:sh.n], sh.master.float())                      # the start values were rounded once: engine == optimizer
    before = sh.master.clone()
    losses = [trainer.train_step(_batch(devt)).item() for _ in range(6)]
    print(f"[emu] LoRA under adamw_bf16: {[round(x, 4) for x in losses]}")
    assert losses[-1] < losses[0] and not torch.equal(before, sh.master)
    assert torch.equal(model.lora_flat[:sh.n], sh.master.float())                      # mirrored after every step
    assert trainer.optimizer._launches == 6                                            # ONE fused launch per step over the bf16 arena
    assert all(p.grad is None for p in model.trainable_parameters())
    st = trainer.optimizer.state[sh.params[0]]
    assert st["step"] == 6.0 and st["exp_avg"].dtype == torch.bfloat16 and float(st["exp_avg"].float().abs().sum()) > 0


def test_lora_under_adamw_bf16_resumes_with_the_optimizer_following_the_loaded_weights(monkeypatch, tmp_path):
    """save_state / load_state with adamw_bf16 over the adapter arena: the fresh trainer's bf16 arena (what the optimizer steps) must follow the LOADED adapter weights —
    otherwise the first resumed step would mirror its own stale start values over them — and the resumed run must continue bit for bit like the uninterrupted one"""
    plug_a, tr_a, cpu, devt = _build(monkeypatch, 1, 1, 2, 16, 16, 32, lora_rank=8, lora_init_b_std=0.02, learning_rate=2e-3, optimizer="adamw_bf16")
    for _ in range(2):
        tr_a.train_step(_batch(devt))
    ck = tmp_path / "checkpoint-2"
    tr_a.save_state(str(ck))
    plug_b, tr_b, _, _ = _build(monkeypatch, 1, 1, 2, 16, 16, 32, lora_rank=8, lora_init_b_std=0.05, learning_rate=2e-3, optimizer="adamw_bf16")   # other start values
    assert not torch.equal(tr_b._bf16_shadow.master, tr_a._bf16_shadow.master)
    tr_b.load_state(str(ck))
    sa, sb = tr_a._bf16_shadow, tr_b._bf16_shadow
    assert torch.equal(sb.master, sa.master) and torch.equal(sb.flat32, sa.flat32) and torch.equal(sb.flat32, sb.master.float())
    la, lb = tr_a.train_step(_batch(devt)), tr_b.train_step(_batch(devt))
    assert torch.equal(la, lb) and torch.equal(sb.master, sa.master)          # same loss from the loaded weights, same bf16 weights after the resumed step
    assert tr_b.state["global_step"] == 3
