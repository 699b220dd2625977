"""UNet (SDXL-style UNet2DConditionModel) on the HIP path vs the fp32 oracle restatement (oracle/unet.py) on identical weights, noised latents,
timesteps and conditioning: prediction, and — for the full fine-tune — the gradient of EVERY parameter tensor vs fp32 autograd.
PARITY UNPINNED against the reference (no golden tensors exist for this network; diffusers is un-vendored): tolerances are stated here —
bf16 HIP vs fp32 oracle: prediction rel-L2 <= 2e-2 / cosine >= 0.9995; per-tensor gradient rel-L2 <= 6e-2 (bias / norm rows <= 8e-2)."""
import pytest
import torch

from oracle.unet import UNetConfig, unet_forward

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16

SMALL = dict(block_out_channels=(64, 128), layers_per_block=1, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
             up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2), attention_head_dim=(1, 2), cross_attention_dim=128,
             projection_class_embeddings_input_dim=64 + 6 * 64, addition_time_embed_dim=64)


# SD1.5-style variants: conv proj_in/out, no addition embedding, heads narrower (40, 80 -> zero-padded flash) and wider (160 -> unfused) than the
# flash kernels' widths
SD15_NARROW = dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                   up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D"), transformer_layers_per_block=(1, 1), attention_head_dim=(8, 8),
                   cross_attention_dim=128, use_linear_projection=False, addition_embed_type=None)
SD15_WIDE = dict(block_out_channels=(64, 320), layers_per_block=1, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
                 up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 1), attention_head_dim=(1, 2),
                 cross_attention_dim=128, use_linear_projection=False, addition_embed_type=None)


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _inputs(B, H, W, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(B, 4, H, W, generator=g).to(BF16)
    t = torch.tensor([17.0, 801.0, 333.0, 950.0][:B])
    ehs = torch.randn(B, 9, 128, generator=g).to(BF16)
    te = torch.randn(B, 64, generator=g).to(BF16)
    ti = torch.tensor([[64.0, 48.0, 0.0, 0.0, 64.0, 48.0]] * B).to(BF16)
    return sample, t, ehs, te, ti


def test_unet_forward_matches_oracle():
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    dev = "cuda:0"
    m = UNet2DConditionModel(device=dev, **SMALL)
    m.init_synthetic(3)
    P = {k: v.float().cpu() for k, v in m.diffusers_state_dict().items()}
    sample, t, ehs, te, ti = _inputs(2, 16, 24, dev)
    out = m(sample.to(dev), t.to(dev), ehs.to(dev), None, added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": ti.to(dev)}, return_dict=False)[0]
    ref = unet_forward(P, UNetConfig(**SMALL), sample.float(), t, ehs.float(), {"text_embeds": te.float(), "time_ids": ti.float()})
    r = _rel(out.cpu(), ref)
    cos = torch.nn.functional.cosine_similarity(out.float().cpu().flatten(), ref.flatten(), dim=0).item()
    print(f"[unet fwd] rel-L2 {r:.3e} cos {cos:.6f}")
    assert out.shape == ref.shape and r < 2e-2 and cos > 0.9995


@pytest.mark.parametrize("arch", ["sdxl_small", "sd15_narrow_heads", "sd15_wide_heads"])
def test_unet_full_finetune_gradients_match_autograd(arch):
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    dev = "cuda:0"
    SMALL = {"sdxl_small": globals()["SMALL"], "sd15_narrow_heads": SD15_NARROW, "sd15_wide_heads": SD15_WIDE}[arch]
    m = UNet2DConditionModel(device=dev, **SMALL)
    m.init_synthetic(5)
    m.enable_full_finetune()
    sd = m.diffusers_state_dict()
    P = {k: v.float().cpu().requires_grad_(True) for k, v in sd.items()}
    sample, t, ehs, te, ti = _inputs(2, 16, 16, dev, seed=1)
    target = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(9))
    ack = {"text_embeds": te.to(dev), "time_ids": ti.to(dev)} if arch == "sdxl_small" else None
    out = m(sample.to(dev), t.to(dev), ehs.to(dev), None, added_cond_kwargs=ack, return_dict=False)[0]
    loss = ((out.float() - target.to(dev)) ** 2).mean()
    loss.backward()
    ref = unet_forward(P, UNetConfig(**SMALL), sample.float(), t, ehs.float(), {"text_embeds": te.float(), "time_ids": ti.float()})
    assert _rel(out.detach().cpu(), ref.detach()) < 2e-2, _rel(out.detach().cpu(), ref.detach())
    lref = ((ref - target) ** 2).mean()
    lref.backward()
    assert abs(loss.item() - lref.item()) < 1e-3 * max(1.0, abs(lref.item())), (loss.item(), lref.item())
    # oracle gradients -> the native layouts (through the same converter that loads checkpoints), then slot by slot
    g = UNet2DConditionModel(device=dev, **SMALL)
    g.load_diffusers_state({k: v.grad for k, v in P.items()})
    worst = (0.0, "")
    for s, sg in zip(m._specs, g._specs):
        got, want = s.g.float().cpu(), sg.t.float().cpu()
        if s.name.startswith("conv_in.weight"):
            got = got[:, :72].reshape(-1, 9, 8)[:, :, :4]; want = want[:, :72].reshape(-1, 9, 8)[:, :, :4]
        if s.name.startswith("conv_out"):
            got, want = got[:4], want[:4]
        r = _rel(got, want)
        tol = 8e-2 if s.kind != "w" else 6e-2
        if r / tol > worst[0]:
            worst = (r / tol, f"{s.name}: {r:.3e}")
        assert r < tol, (s.name, r)
    print(f"[unet grads {arch}] {len(m._specs)} tensors, worst (relative to its tolerance) {worst[1]}")


def test_unet_lora_gradients_match_autograd():
    """frozen base + peft-style LoRA (y = W x + (alpha/r) B A x) on every attn1 / attn2 projection: prediction and every adapter gradient vs autograd"""
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    dev = "cuda:0"
    rank, alpha = 16, 16.0
    m = UNet2DConditionModel(device=dev, **SMALL)
    m.init_synthetic(6)
    params = m.add_lora_adapter(rank=rank, alpha=alpha, seed=3, init_b_std=0.05)
    P = {k: v.float().cpu() for k, v in m.diffusers_state_dict().items()}
    lora = {n: p.detach().float().cpu().requires_grad_(True) for n, p in m.named_parameters() if ".lora_" in n}
    assert len(lora) == len(params) == 2 * 8 * 8 and all(p.requires_grad for p in params)        # 8 transformer layers x (q,k,v,out) x 2 attentions x (A,B)
    # oracle: merge the adapters into effective weights (exactly what the fused K-extension computes), autograd through the merge
    Pe = dict(P)
    for n in lora:
        if ".lora_A." in n:
            base = n.replace(".lora_A.default.weight", "")
            Pe[base + ".weight"] = P[base + ".weight"] + (alpha / rank) * lora[base + ".lora_B.default.weight"] @ lora[n]
    sample, t, ehs, te, ti = _inputs(2, 16, 16, dev, seed=2)
    target = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(8))
    out = m(sample.to(dev), t.to(dev), ehs.to(dev), None, added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": ti.to(dev)}, return_dict=False)[0]
    loss = ((out.float() - target.to(dev)) ** 2).mean()
    loss.backward()
    ref = unet_forward(Pe, UNetConfig(**SMALL), sample.float(), t, ehs.float(), {"text_embeds": te.float(), "time_ids": ti.float()})
    assert _rel(out.detach().cpu(), ref.detach()) < 2e-2
    ((ref - target) ** 2).mean().backward()
    worst = (0.0, "")
    for n, p in m.named_parameters():
        if ".lora_" not in n:
            assert p.grad is None
            continue
        r = _rel(p.grad.cpu(), lora[n].grad)
        if r > worst[0]:
            worst = (r, n)
        assert r < 6e-2, (n, r)
    print(f"[unet lora grads] {len(lora)} tensors, worst {worst[1]}: {worst[0]:.3e}")


def test_lora_checkpoint_roundtrip_on_device(tmp_path):
    """save_lora_weights -> wipe the adapters -> load_lora_weights: the adapter tensors (views into the K-extended weights) and the prediction come back
    bit-identical; file layout = pytorch_lora_weights.safetensors with `unet.<module>.lora_{A,B}.weight` keys"""
    from types import SimpleNamespace

    from safetensors.torch import load_file

    from simpletuner_amd.foundation import ModelFoundation
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    dev = "cuda:0"

    class Plug(ModelFoundation):
        MODEL_SUBFOLDER = "unet"

    m = UNet2DConditionModel(device=dev, **SMALL)
    m.init_synthetic(6)
    m.add_lora_adapter(rank=8, alpha=8.0, seed=4, init_b_std=0.05)
    plug = Plug(SimpleNamespace(), SimpleNamespace(device=torch.device(dev)))
    plug.model = m
    sample, t, ehs, te, ti = _inputs(2, 16, 16, dev, seed=2)
    ack = {"text_embeds": te.to(dev), "time_ids": ti.to(dev)}
    with torch.no_grad():
        before = m(sample.to(dev), t.to(dev), ehs.to(dev), None, added_cond_kwargs=ack, return_dict=False)[0].clone()
    path = plug.save_lora_weights(str(tmp_path))
    flat = load_file(path)
    assert len(flat) == 2 * 8 * 8 and all(k.startswith("unet.") and (".lora_A.weight" in k or ".lora_B.weight" in k) for k in flat)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if ".lora_" in n:
                p.zero_()
        wiped = m(sample.to(dev), t.to(dev), ehs.to(dev), None, added_cond_kwargs=ack, return_dict=False)[0].clone()
    assert not torch.equal(wiped, before)
    plug.load_lora_weights(input_dir=str(tmp_path))
    with torch.no_grad():
        after = m(sample.to(dev), t.to(dev), ehs.to(dev), None, added_cond_kwargs=ack, return_dict=False)[0]
    assert torch.equal(after, before)


def test_min_snr_weighted_loss_matches_formula():
    """DDPM epsilon loss with snr_gamma (common.py:6363-6397): mean_b( w_b * mean_chw (pred - noise)^2 ), w = min(snr, gamma)/snr, + its gradient"""
    from simpletuner_amd.sdxl.model import SDXL
    from simpletuner_amd.training.trainer import St355Accelerator, default_config
    dev = torch.device("cuda", 0)
    cfg = default_config(model_family="sdxl", model_type="full", snr_gamma=5.0)
    pl = SDXL(cfg, St355Accelerator(dev))
    pl.setup_training_noise_schedule()
    g = torch.Generator().manual_seed(0)
    pred = torch.randn(3, 4, 16, 16, generator=g).to(BF16).to(dev).requires_grad_(True)
    noise = torch.randn(3, 4, 16, 16, generator=g).to(BF16).to(dev)
    t = torch.tensor([5, 400, 990], device=dev)
    loss = pl.loss({"noise": noise, "timesteps": t}, {"model_prediction": pred})
    loss.backward()
    snr = pl.noise_schedule.snr(t.cpu())
    w = torch.minimum(snr, torch.tensor(5.0)) / snr
    pf = pred.detach().float().cpu().requires_grad_(True)
    ref = (((pf - noise.float().cpu()) ** 2).mean(dim=(1, 2, 3)) * w).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-4 * max(1.0, ref.item())
    assert _rel(pred.grad.cpu(), pf.grad) < 1e-2


def test_ddpm_prepare_batch_offset_noise_and_input_perturbation():
    """common.py:5940-5968: the offset is part of `noise` (the epsilon target), the perturbation only of `input_noise`; x_t is built from input_noise"""
    from simpletuner_amd.sdxl.model import SDXL
    from simpletuner_amd.training.trainer import St355Accelerator, default_config
    dev = torch.device("cuda", 0)
    cfg = default_config(model_family="sdxl", model_type="full", offset_noise=True, noise_offset=0.1, noise_offset_probability=1.0, input_perturbation=0.1)
    pl = SDXL(cfg, St355Accelerator(dev))
    pl.setup_training_noise_schedule()
    torch.manual_seed(3)
    lat = torch.randn(2, 4, 16, 16, device=dev).to(BF16)
    t = torch.tensor([10, 900])
    out = pl.prepare_batch({"latent_batch": lat, "prompt_embeds": torch.zeros(2, 9, 128, device=dev, dtype=BF16), "timesteps": t}, {"global_step": 0})
    n, n_in = out["noise"].float(), out["input_noise"].float()
    assert not torch.equal(n, n_in) and (n_in - n).std().item() == pytest.approx(0.1, rel=0.2)
    assert pl.get_prediction_target(out) is out["noise"]
    a, b = pl.noise_schedule.mix_coefficients(t.to(dev))
    ref = a.view(-1, 1, 1, 1) * lat.float() + b.view(-1, 1, 1, 1) * n_in
    assert _rel(out["noisy_latents"], ref) < 5e-3


@pytest.mark.parametrize("mode,interval,stride,lora", [("layer", None, None, False), ("layer", None, None, True), ("every_second_unit", 2, None, False)])
def test_unet_checkpointed_gradients_equal_direct_gradients(mode, interval, stride, lora):
    """SURVEY.md §8(f)3 for the UNet families (diffusers' `enable_gradient_checkpointing`: every ResnetBlock2D / transformer is its own checkpoint — the `layer`
    rows of documentation/experimental/SEGMENTED_CHECKPOINTING.md:774-776, 834-836): a checkpointed unit runs without a tape and is re-run on a private tape in
    backward — prediction and the gradient arena are BIT-identical to the run that records everything, with fewer activations alive between forward and backward"""
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    dev = "cuda:0"

    def run(ckpt):
        import gc
        gc.collect(); torch.cuda.empty_cache()
        m = UNet2DConditionModel(device=dev, **SMALL)
        m.init_synthetic(5)
        if lora:
            m.add_lora_adapter(rank=8, alpha=8.0, seed=4, init_b_std=0.05)
        else:
            m.enable_full_finetune()
        if ckpt:
            m.enable_gradient_checkpointing()
            m.set_gradient_checkpointing_interval(interval)
            m.set_gradient_checkpointing_segment_stride(stride)
        sample, t, ehs, te, ti = _inputs(2, 16, 16, dev, seed=1)
        target = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(9)).to(dev)
        args = (sample.to(dev), t.to(dev), ehs.to(dev), None)
        ack = {"text_embeds": te.to(dev), "time_ids": ti.to(dev)}
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        out = m(*args, added_cond_kwargs=ack, return_dict=False)[0]
        loss = ((out.float() - target) ** 2).mean()
        kept = torch.cuda.memory_allocated() - base                    # what the forward left alive for the backward
        loss.backward()
        torch.cuda.synchronize()
        return out.detach().clone(), torch.cat([p.grad.detach().reshape(-1).float() for p in m.trainable_parameters()]).clone(), kept, m._unit_counter
    o0, g0, kept0, _ = run(False)
    o1, g1, kept1, units = run(True)
    assert torch.equal(o0, o1) and torch.equal(g0, g1) and g0.abs().sum().item() > 0
    print(f"[ckpt unet {'lora' if lora else 'full'}] {mode}: {units} units, memory held between forward and backward {kept0 / 2**20:.1f} MiB -> {kept1 / 2**20:.1f} MiB")
    assert kept1 < kept0


def test_sdxl_ddim_cfg_sampling_trajectory_matches_oracle_forward():
    """§8(f)4 for the epsilon families: DDIM (the reference's DEFAULT_NOISE_SCHEDULER for SDXL / SD1.x / PixArt) + classifier-free guidance through the SDXL
    plugin's own forward (time ids / pooled text embeds paired [negative ; positive]) against the same loop driven by the fp32 oracle UNet: 4 steps"""
    from types import SimpleNamespace
    from simpletuner_amd.sampling import DDIMScheduler, cfg_combine, sample_images
    from simpletuner_amd.sdxl.model import SDXL
    from simpletuner_amd.training.trainer import St355Accelerator, default_config
    dev = torch.device("cuda:0")
    cfg = default_config(model_family="sdxl", model_type="lora", train_batch_size=2)
    pl = SDXL(cfg, St355Accelerator(dev))
    torch.manual_seed(0)
    pl.load_model(**SMALL)
    m = pl.get_trained_component()
    P = {k: v.float().cpu() for k, v in m.diffusers_state_dict().items()}
    sample, _, ehs, te, ti = _inputs(2, 16, 16, dev, seed=4)
    g = torch.Generator().manual_seed(5)
    neg_e, neg_te = torch.randn(2, 9, 128, generator=g).to(BF16), torch.randn(2, 64, generator=g).to(BF16)
    gs = 5.0
    with torch.no_grad():
        out = sample_images(pl, ehs, te, 16, 16, num_inference_steps=4, decode=False, guidance_scale=gs, negative_prompt_embeds=neg_e, negative_pooled=neg_te,
                            latents=sample.clone(), extra_batch={"added_cond_kwargs": {"time_ids": ti.to(dev)}})
    sc = DDIMScheduler(timestep_spacing="trailing")      # validation.py:2889-2892: the trainer default the sampler now reads
    sc.set_timesteps(4)
    x = sample.float()
    e2, t2, i2 = torch.cat([neg_e.float(), ehs.float()]), torch.cat([neg_te.float(), te.float()]), torch.cat([ti.float(), ti.float()])
    for t in sc.timesteps:
        pred = unet_forward(P, UNetConfig(**SMALL), torch.cat([x, x]), t.float().expand(4), e2, {"text_embeds": t2, "time_ids": i2})
        x = sc.step(cfg_combine(pred, gs), t, x, return_dict=False)[0]
    r = _rel(out.cpu(), x)
    print(f"[sampling sdxl] DDIM 4 steps, CFG {gs}: final latents HIP vs oracle-driven loop rel-L2 {r:.3e}")
    assert torch.isfinite(out.float()).all() and r < 3e-2
