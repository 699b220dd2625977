"""CPU-side checks of the UNet path's host logic: the DDPM training schedule (diffusers DDPMScheduler arithmetic as configured for SD1.5 / SDXL:
scaled_linear betas 0.00085 -> 0.012, 1000 steps), the segmented timestep selection (custom_schedule.py:18-58), the oracle UNet's structure
(state-dict round trip through the native layouts) and its FLOP counter against the figures SURVEY.md §8(d) quotes."""
import math

import pytest
import torch

from oracle.unet import UNetConfig, unet_flops_fwd, unet_forward
from simpletuner_amd.foundation import DDPMSchedule


def test_ddpm_schedule_known_values():
    s = DDPMSchedule()
    acp = s.alphas_cumprod
    assert acp.shape == (1000,)
    assert abs(acp[0].item() - (1 - 0.00085)) < 1e-7
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
    ref = torch.cumprod(1 - betas, 0)
    assert torch.allclose(acp.double(), ref, atol=2e-6)
    assert abs(acp[999].item() - 0.0046600) < 2e-5                 # the well-known terminal SNR of the SD schedule
    a, b = s.mix_coefficients(torch.tensor([0, 500, 999]))
    assert torch.allclose(a * a + b * b, torch.ones(3), atol=1e-6)


def test_segmented_timestep_selection_covers_one_segment_each():
    s = DDPMSchedule()
    torch.manual_seed(0)
    for bsz in (2, 4, 7):
        seg = 1000 // bsz
        for _ in range(20):
            t = s.sample_timesteps(bsz)
            assert t.shape == (bsz,) and t.dtype == torch.long
            for i in range(bsz):
                start = 999 - i * seg
                end = max(start - seg, 0) if i != bsz - 1 else 0
                assert end <= int(t[i]) <= start
    assert s.sample_timesteps(1).shape == (1,)


def test_unet_flop_counter_matches_published_figures():
    assert abs(unet_flops_fwd(UNetConfig(), 128, 128) / 1e12 - 6.76) < 0.05        # SDXL @1024^2: "~6.0 TFLOP" published, 6.76 counted (incl. attention)
    assert abs(unet_flops_fwd(UNetConfig.sd15(), 64, 64) / 1e12 - 0.80) < 0.02      # SD1.5 @512^2: 0.8 TFLOP


def test_oracle_unet_runs_and_is_conditioned():
    cfg = UNetConfig(block_out_channels=(32, 64), layers_per_block=1, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
                     up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 1), attention_head_dim=(1, 2), cross_attention_dim=32,
                     projection_class_embeddings_input_dim=16 + 6 * 8, addition_time_embed_dim=8, norm_num_groups=8)
    g = torch.Generator().manual_seed(0)
    P = {}

    def lin(n, o, i, bias=True):
        P[n + ".weight"] = torch.randn(o, i, generator=g) / math.sqrt(i)
        if bias:
            P[n + ".bias"] = torch.randn(o, generator=g) * 0.02

    def conv(n, o, i, k=3):
        P[n + ".weight"] = torch.randn(o, i, k, k, generator=g) / math.sqrt(i * k * k)
        P[n + ".bias"] = torch.randn(o, generator=g) * 0.02

    def norm(n, c):
        P[n + ".weight"] = torch.ones(c); P[n + ".bias"] = torch.zeros(c)

    def res(p, i, o):
        norm(p + "norm1", i); conv(p + "conv1", o, i); lin(p + "time_emb_proj", o, 128); norm(p + "norm2", o); conv(p + "conv2", o, o)
        if i != o:
            conv(p + "conv_shortcut", o, i, 1)

    def tr(p, c):
        norm(p + "norm", c); lin(p + "proj_in", c, c); lin(p + "proj_out", c, c)
        q = p + "transformer_blocks.0."
        for nm in ("norm1", "norm2", "norm3"):
            norm(q + nm, c)
        for a, kd in (("attn1.", c), ("attn2.", 32)):
            lin(q + a + "to_q", c, c, False); lin(q + a + "to_k", c, kd, False); lin(q + a + "to_v", c, kd, False); lin(q + a + "to_out.0", c, c)
        lin(q + "ff.net.0.proj", 8 * c, c); lin(q + "ff.net.2", c, 4 * c)

    conv("conv_in", 32, 4); lin("time_embedding.linear_1", 128, 32); lin("time_embedding.linear_2", 128, 128)
    lin("add_embedding.linear_1", 128, 64); lin("add_embedding.linear_2", 128, 128)
    res("down_blocks.0.resnets.0.", 32, 32); conv("down_blocks.0.downsamplers.0.conv", 32, 32)
    res("down_blocks.1.resnets.0.", 32, 64); tr("down_blocks.1.attentions.0.", 64)
    res("mid_block.resnets.0.", 64, 64); tr("mid_block.attentions.0.", 64); res("mid_block.resnets.1.", 64, 64)
    res("up_blocks.0.resnets.0.", 128, 64); tr("up_blocks.0.attentions.0.", 64); res("up_blocks.0.resnets.1.", 96, 64); tr("up_blocks.0.attentions.1.", 64)
    conv("up_blocks.0.upsamplers.0.conv", 64, 64)
    res("up_blocks.1.resnets.0.", 96, 32); res("up_blocks.1.resnets.1.", 64, 32)
    norm("conv_norm_out", 32); conv("conv_out", 4, 32)
    x = torch.randn(2, 4, 8, 8, generator=g)
    ctx = torch.randn(2, 5, 32, generator=g)
    add = {"text_embeds": torch.randn(2, 16, generator=g), "time_ids": torch.tensor([[64.0, 64, 0, 0, 64, 64]] * 2)}
    y1 = unet_forward(P, cfg, x, torch.tensor([10.0, 900.0]), ctx, add)
    y2 = unet_forward(P, cfg, x, torch.tensor([500.0, 900.0]), ctx, add)
    y3 = unet_forward(P, cfg, x, torch.tensor([10.0, 900.0]), ctx * 0.5, add)
    assert y1.shape == (2, 4, 8, 8) and torch.isfinite(y1).all()
    assert not torch.allclose(y1[0], y2[0]) and torch.allclose(y1[1], y2[1], atol=1e-5)       # timestep conditioning is per sample
    assert not torch.allclose(y1, y3)                                                           # cross-attention sees the text tokens


def test_min_snr_weights_match_oracle_formula():
    """DDPMSchedule.min_snr_weights == min(snr, gamma) / snr (epsilon) or / (snr + 1) (v-prediction) with snr from the oracle's compute_snr"""
    from oracle import train_math as TM
    s = DDPMSchedule()
    t = torch.tensor([0, 13, 500, 998, 999])
    snr = TM.compute_snr(t, s.alphas_cumprod)
    assert torch.allclose(s.snr(t), snr, rtol=1e-5)
    for gamma in (1.0, 5.0):
        assert torch.allclose(s.min_snr_weights(t, gamma, False), torch.minimum(snr, torch.tensor(gamma)) / snr, rtol=1e-5)
        assert torch.allclose(s.min_snr_weights(t, gamma, True), torch.minimum(snr, torch.tensor(gamma)) / (snr + 1), rtol=1e-5)


def test_tape_reverse_sweep_sums_fanout_gradients():
    """the UNet engine's reverse tape (simpletuner_amd/unet/unet.py::Tape) vs autograd on a tiny graph with a skip connection and a shared input
    (the two places a UNet sums gradients: up-block concatenation skips, the time embedding feeding every ResNet)."""
    import simpletuner_amd.ops as ops
    from simpletuner_amd.unet.unet import Tape
    real_add = ops.add
    ops.add = lambda a, b: a + b                      # the tape sums meeting gradients with ops.add (a HIP launch on device tensors)
    try:
        x = torch.randn(4, 3, dtype=torch.float64, requires_grad=True)
        e = torch.randn(4, 3, dtype=torch.float64, requires_grad=True)
        # reference
        h1 = x * 2 + e
        h2 = torch.tanh(h1) * e
        y = torch.cat([h2, h1], dim=1).sum(dim=1, keepdim=True) * h2
        g = torch.randn_like(y)
        y.backward(g)
        # same graph on the tape
        T = Tape()
        xd, ed = x.detach(), e.detach()
        grads = {}
        T.rec([xd], [None], lambda d: (grads.__setitem__("x", d) or None,))     # leaf sentinels (first on the tape = last in the sweep)
        T.rec([ed], [None], lambda d: (grads.__setitem__("e", d) or None,))
        a1 = xd * 2 + ed
        T.rec([a1], [xd, ed], lambda d: (2 * d, d))
        a2 = torch.tanh(a1) * ed
        T.rec([a2], [a1, ed], lambda d: (d * ed * (1 - torch.tanh(a1) ** 2), d * torch.tanh(a1)))
        c = torch.cat([a2, a1], dim=1)
        T.rec([c], [a2, a1], lambda d: (d[:, :3], d[:, 3:]))
        s = c.sum(dim=1, keepdim=True)
        T.rec([s], [c], lambda d: (d.expand(-1, 6),))
        out = s * a2
        T.rec([out], [s, a2], lambda d: ((d * a2).sum(dim=1, keepdim=True), d * s))
        T.backward(out, g)
        assert torch.allclose(grads["x"], x.grad) and torch.allclose(grads["e"], e.grad)
    finally:
        ops.add = real_add


def test_ddpm_sampling_matches_reference_code_outputs():
    """DDPMSchedule.timestep_weights / sample_timesteps vs the outputs of the reference's generate_timestep_weights and
    segmented_timestep_selection executed with a pinned torch RNG (tools/gen_golden.py::gen_ddpm_sampling -> tests/golden/ddpm_sampling_vectors.pt)"""
    from pathlib import Path
    from types import SimpleNamespace
    G = torch.load(Path(__file__).parent / "golden" / "ddpm_sampling_vectors.pt")
    for strat, (args, want) in G["weights"].items():
        got = DDPMSchedule.timestep_weights(SimpleNamespace(**args), 1000)
        assert torch.equal(got, want), strat
    s = DDPMSchedule()
    for bsz, seed, want in G["segmented"]:
        torch.manual_seed(seed)
        assert torch.equal(s.sample_timesteps(bsz), want), (bsz, seed)


def test_lora_checkpoint_layout_roundtrip(tmp_path):
    """save_lora_weights / load_lora_weights: `pytorch_lora_weights.safetensors`, keys `<subfolder>.<module>.lora_A.weight` (peft state-dict layout under
    diffusers' component prefix — what save_hooks.py:850-895 hands to the pipeline's save_lora_weights)"""
    from types import SimpleNamespace

    from safetensors.torch import load_file

    from simpletuner_amd.foundation import ModelFoundation

    class Comp(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleDict({"0": torch.nn.ModuleDict({"to_q": torch.nn.ModuleDict({"lora_A": torch.nn.ModuleDict({"default": torch.nn.Linear(8, 2, bias=False)}),
                                                                                                          "lora_B": torch.nn.ModuleDict({"default": torch.nn.Linear(2, 8, bias=False)})})})})
            self.base = torch.nn.Linear(8, 8)

    class Plug(ModelFoundation):
        MODEL_SUBFOLDER = "unet"

    m = Plug(SimpleNamespace(), SimpleNamespace(device=torch.device("cpu")))
    m.model = Comp()
    path = m.save_lora_weights(str(tmp_path))
    flat = load_file(path)
    assert sorted(flat) == ["unet.blocks.0.to_q.lora_A.weight", "unet.blocks.0.to_q.lora_B.weight"]
    want = {k: v.clone() for k, v in flat.items()}
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            if ".lora_" in n:
                p.add_(1.0)
    m.load_lora_weights(input_dir=str(tmp_path))
    assert torch.equal(m.model.blocks["0"]["to_q"]["lora_A"]["default"].weight, want["unet.blocks.0.to_q.lora_A.weight"])
    assert torch.equal(m.model.blocks["0"]["to_q"]["lora_B"]["default"].weight, want["unet.blocks.0.to_q.lora_B.weight"])


def test_dit_plugins_refuse_tokenwise_timesteps_and_reference_tokens():
    """the reference's Kontext inputs (clean conditioning tokens, tests/test_flux_model.py:243-272) are refused loudly, never silently mis-conditioned; tokenwise
    timesteps are taken by all three DiT plugins since round 4 (the PixArt ControlNet wrapper refuses them, as the reference's does not handle them)"""
    from types import SimpleNamespace

    import pytest

    from simpletuner_amd.flux.model import Flux
    from simpletuner_amd.pixart.model import PixartSigma
    from simpletuner_amd.sd3.model import SD3
    for cls in (Flux, SD3, PixartSigma):
        m = cls.__new__(cls)
        m.config, m.accelerator = SimpleNamespace(), SimpleNamespace(device=torch.device("cpu"))
        if cls is PixartSigma:
            with pytest.raises(NotImplementedError, match="tokenwise timesteps"):
                m._controlnet_predict_single({"timesteps": torch.tensor([[100.0, 900.0, 500.0, 700.0]]), "latents": torch.zeros(1, 4, 4, 4)})
        if cls is not Flux:              # Flux appends them at t = 0 (test_flux_plugin_appends_clean_conditioning_tokens_at_t_zero)
            with pytest.raises(NotImplementedError, match="conditioning_packed_latents"):
                m._model_predict_single({"timesteps": torch.tensor([100.0]), "latents": torch.zeros(1, 16, 4, 4), "conditioning_packed_latents": torch.zeros(1, 2, 64)})


def test_sd3_plugin_hands_tokenwise_timesteps_to_the_transformer():
    """the reference's tests/test_sd3_model.py:179-204 (test_model_predict_accepts_tokenwise_timesteps) on the st355 plugin: [B, S_img] timesteps reach the
    transformer's `timestep` argument unchanged (0..1000 scale, fp32)"""
    from types import SimpleNamespace

    from simpletuner_amd.sd3.model import SD3
    m = SD3.__new__(SD3)
    m.config, m.accelerator = SimpleNamespace(), SimpleNamespace(device=torch.device("cpu"))
    seen = {}

    def fake(**kw):
        seen.update(kw)
        return (torch.randn(1, 16, 4, 4),)

    m.model = fake
    batch = {"noisy_latents": torch.randn(1, 16, 4, 4), "timesteps": torch.tensor([[100.0, 900.0, 500.0, 700.0]]), "encoder_hidden_states": torch.randn(1, 3, 64),
             "add_text_embeds": torch.randn(1, 32)}
    out = m._model_predict_single(batch)
    assert out["model_prediction"].shape == (1, 16, 4, 4)
    assert seen["timestep"].dtype == torch.float32 and torch.equal(seen["timestep"], batch["timesteps"])


def test_pixart_plugin_hands_tokenwise_timesteps_to_the_transformer():
    """the reference's tests/test_pixart_model.py:91-115 on the st355 plugin: [B, S] timesteps reach the transformer unchanged"""
    from types import SimpleNamespace

    from simpletuner_amd.pixart.model import PixartSigma
    m = PixartSigma.__new__(PixartSigma)
    m.config, m.accelerator = SimpleNamespace(), SimpleNamespace(device=torch.device("cpu"))
    seen = {}

    def fake(*a, **kw):
        seen.update(kw)
        return (torch.randn(1, 8, 4, 4),)

    m.model = fake
    batch = {"noisy_latents": torch.randn(1, 4, 4, 4), "timesteps": torch.tensor([[100, 900, 200, 800]]), "encoder_hidden_states": torch.randn(1, 4, 16),
             "encoder_attention_mask": torch.ones(1, 4), "resolution": torch.tensor([[4.0, 4.0]]), "aspect_ratio": torch.tensor([[1.0]])}
    out = m._model_predict_single(batch)
    assert out["model_prediction"].shape == (1, 4, 4, 4) and torch.equal(seen["timestep"], batch["timesteps"])


def test_flux_plugin_hands_tokenwise_timesteps_to_the_transformer():
    """the reference's tests/test_flux_model.py:213-241 (test_model_predict_accepts_tokenwise_timesteps) on the st355 plugin: [B, S_img] timesteps reach the
    transformer divided by 1000 (the transformer multiplies them back, flux/model.py:739-745)"""
    from types import SimpleNamespace

    from simpletuner_amd.flux.model import Flux
    m = Flux.__new__(Flux)
    m.config, m.accelerator, m._ids_cache = SimpleNamespace(), SimpleNamespace(device=torch.device("cpu")), {}
    seen = {}

    def fake(**kw):
        seen.update(kw)
        return (torch.randn(1, 4, 64).to(torch.bfloat16),)

    m.model = fake
    m.get_trained_component = lambda: SimpleNamespace(config=SimpleNamespace(guidance_embeds=False))
    import simpletuner_amd.flux.model as FM
    orig, orig_pack = FM._UnpackFn.apply, FM.pack_latents
    FM._UnpackFn.apply = staticmethod(lambda packed, h, w: torch.zeros(packed.shape[0], 16, h // 8, w // 8))
    FM.pack_latents = lambda x: x.reshape(x.shape[0], -1, 64)             # (the device pack kernel is not what this test is about)
    try:
        batch = {"noisy_latents": torch.randn(1, 16, 4, 4), "latents": torch.randn(1, 16, 4, 4), "timesteps": torch.tensor([[100.0, 900.0, 500.0, 700.0]]),
                 "prompt_embeds": torch.randn(1, 3, 16), "add_text_embeds": torch.randn(1, 8)}
        out = m._model_predict_single(batch)
        assert out["model_prediction"].shape == (1, 16, 4, 4)
        assert torch.allclose(seen["timestep"], torch.tensor([[0.1, 0.9, 0.5, 0.7]]))
        with pytest.raises(ValueError, match="sequence length"):
            m._model_predict_single(dict(batch, timesteps=torch.tensor([[100.0, 900.0, 500.0]])))
    finally:
        FM._UnpackFn.apply, FM.pack_latents = orig, orig_pack


def test_flux_plugin_appends_clean_conditioning_tokens_at_t_zero():
    """the reference's tests/test_flux_model.py:243-272 (test_model_predict_appends_clean_conditioning_timesteps) on the st355 plugin: Kontext's packed reference-image
    tokens are appended to the scene tokens (and their ids to the image ids), conditioned on t = 0 — timesteps [[0.1, 0.9, 0.5, 0.7, 0.0, 0.0]] — and dropped from
    the prediction before unpacking"""
    from types import SimpleNamespace

    import simpletuner_amd.flux.model as FM
    m = FM.Flux.__new__(FM.Flux)
    m.config, m.accelerator, m._ids_cache = SimpleNamespace(model_flavour="kontext"), SimpleNamespace(device=torch.device("cpu")), {}
    seen = {}

    def fake(**kw):
        seen.update(kw)
        return (torch.arange(6 * 64, dtype=torch.float32).view(1, 6, 64).to(torch.bfloat16),)

    m.model = fake
    m.get_trained_component = lambda: SimpleNamespace(config=SimpleNamespace(guidance_embeds=False))
    orig, orig_pack = FM._UnpackFn.apply, FM.pack_latents
    got = {}
    FM._UnpackFn.apply = staticmethod(lambda packed, h, w: got.setdefault("packed", packed) is None or torch.zeros(packed.shape[0], 16, h // 8, w // 8))
    FM.pack_latents = lambda x: x.reshape(x.shape[0], -1, 64)
    try:
        batch = {"noisy_latents": torch.randn(1, 16, 4, 4), "latents": torch.randn(1, 16, 4, 4), "timesteps": torch.tensor([[100.0, 900.0, 500.0, 700.0]]),
                 "prompt_embeds": torch.randn(1, 3, 16), "add_text_embeds": torch.randn(1, 8), "conditioning_packed_latents": torch.randn(1, 2, 64),
                 "conditioning_ids": torch.zeros(1, 2, 3)}
        out = m._model_predict_single(batch)
        assert out["model_prediction"].shape == (1, 16, 4, 4)
        assert torch.allclose(seen["timestep"], torch.tensor([[0.1, 0.9, 0.5, 0.7, 0.0, 0.0]]))
        assert seen["hidden_states"].shape == (1, 6, 64) and seen["img_ids"].shape == (1, 6, 3)
        assert got["packed"].shape == (1, 4, 64)                                    # the two reference-image tokens were dropped
        # per-sample timesteps are broadcast over the scene tokens first (flux/model.py:616-617)
        m._model_predict_single(dict(batch, timesteps=torch.tensor([250.0])))
        assert torch.allclose(seen["timestep"], torch.tensor([[0.25, 0.25, 0.25, 0.25, 0.0, 0.0]]))
    finally:
        FM._UnpackFn.apply, FM.pack_latents = orig, orig_pack


def test_vae_seam_attributes_and_latent_scaling_rule():
    """plugin class attributes the trainer reads (common.py:451-530) and scale_vae_latents_for_cache (foundation_mixins.py:66-79)"""
    from types import SimpleNamespace

    from simpletuner_amd.flux.model import Flux
    from simpletuner_amd.pixart.model import PixartSigma
    from simpletuner_amd.sd1x.model import StableDiffusion1
    from simpletuner_amd.sd3.model import SD3
    from simpletuner_amd.sdxl.model import SDXL
    for cls, ch in ((Flux, 16), (SD3, 16), (SDXL, 4), (StableDiffusion1, 4), (PixartSigma, 4)):
        assert cls.autoencoder_class().__name__ == "AutoencoderKL" and cls.VAE_CONFIG["latent_channels"] == ch == cls.LATENT_CHANNEL_COUNT
        for attr in ("NAME", "PREDICTION_TYPE", "MODEL_TYPE", "MODEL_CLASS", "MODEL_SUBFOLDER", "PIPELINE_CLASSES", "HUGGINGFACE_PATHS", "DEFAULT_MODEL_FLAVOUR",
                     "TEXT_ENCODER_CONFIGURATION", "LATENT_CHANNEL_COUNT", "DEFAULT_LORA_TARGET", "DDP_FIND_UNUSED_PARAMETERS"):
            assert hasattr(cls, attr), (cls.__name__, attr)
        assert cls.DEFAULT_MODEL_FLAVOUR in cls.HUGGINGFACE_PATHS
    m = Flux.__new__(Flux)
    z = torch.tensor([1.0, -2.0])
    shifted = SimpleNamespace(config=SimpleNamespace(shift_factor=0.1159, scaling_factor=0.3611))
    plain = SimpleNamespace(config=SimpleNamespace(shift_factor=None, scaling_factor=0.13025))
    assert torch.allclose(m.scale_vae_latents_for_cache(z, shifted), (z - 0.1159) * 0.3611)
    assert torch.allclose(m.scale_vae_latents_for_cache(z, plain), z * 0.13025)
    assert m.scale_vae_latents_for_cache(z, None) is z and m.scale_vae_latents_for_cache(None, plain) is None
    m.AUTOENCODER_SCALING_FACTOR = 2.0                              # a family-level override wins over the VAE's own factor
    assert torch.allclose(m.scale_vae_latents_for_cache(z, plain), z * 2.0)


def test_zero_terminal_snr_rescale_matches_reference_code_output():
    """rescale_betas_zero_snr: enforce_zero_terminal_snr (custom_schedule.py:157-175) executed by tools/gen_golden.py on the scaled-linear betas"""
    from pathlib import Path

    from simpletuner_amd.foundation import DDPMSchedule, enforce_zero_terminal_snr
    want = torch.load(Path(__file__).parent / "golden" / "ddpm_sampling_vectors.pt", weights_only=False)["zero_terminal_snr_betas"]
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    torch.testing.assert_close(enforce_zero_terminal_snr(betas), want, rtol=0, atol=0)
    s = DDPMSchedule(prediction_type="v_prediction", rescale_betas_zero_snr=True)
    assert s.alphas_cumprod[-1].item() == 0.0 and s.alphas_cumprod[0].item() == pytest.approx(DDPMSchedule().alphas_cumprod[0].item(), rel=1e-6)
    w = s.min_snr_weights(torch.tensor([0, 500, 999]), 5.0, v_prediction=True)
    assert torch.isfinite(w).all() and w[-1].item() == 0.0                      # terminal step: snr 0 -> weight 0 with the v-prediction divisor
    assert DDPMSchedule().config.rescale_betas_zero_snr is False


def test_flop_counters_agree_on_product_and_oracle_configs():
    """bench.py counts FLOPs from the PRODUCT models' config objects (tools/flop_count.py; no oracle import outside its cpu_baseline leg): same
    numbers as on the oracle's dataclasses for SDXL, SD1.5, PixArt-Sigma 2K and the SDXL VAE"""
    import ast
    import inspect
    from pathlib import Path
    from types import SimpleNamespace

    from oracle.pixart import PixArtConfig
    from oracle.vae import VAEConfig
    from simpletuner_amd.sd1x.model import SD15_ARCH
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    from tools.flop_count import pixart_flops_fwd, unet_flops_fwd, vae_encoder_flops

    def product_unet_cfg(**arch):
        d = {k: v.default for k, v in inspect.signature(UNet2DConditionModel.__init__).parameters.items() if v.default is not inspect.Parameter.empty}
        d.update(arch)
        nb = len(d["block_out_channels"])
        if isinstance(d["transformer_layers_per_block"], int):
            d["transformer_layers_per_block"] = (d["transformer_layers_per_block"],) * nb
        return SimpleNamespace(**d)

    assert unet_flops_fwd(product_unet_cfg(), 128, 128, 77) == unet_flops_fwd(UNetConfig(), 128, 128, 77)
    assert unet_flops_fwd(product_unet_cfg(**SD15_ARCH), 64, 64, 77) == unet_flops_fwd(UNetConfig.sd15(), 64, 64, 77)
    pix = SimpleNamespace(num_attention_heads=16, attention_head_dim=72, num_layers=28, patch_size=2, in_channels=4, out_channels=8)
    assert pixart_flops_fwd(pix, 256, 256, 300, 13) == pixart_flops_fwd(PixArtConfig(sample_size=256), 256, 256, 300, 13)
    vae = SimpleNamespace(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2)
    assert vae_encoder_flops(vae, 1024, 1024) == vae_encoder_flops(VAEConfig(), 1024, 1024)
    # bench.py: the only function that imports the oracle is the cpu_baseline leg
    tree = ast.parse((Path(__file__).parent.parent / "bench.py").read_text())
    offenders = {fn.name for fn in ast.walk(tree) if isinstance(fn, ast.FunctionDef) for n in ast.walk(fn)
                 if isinstance(n, (ast.Import, ast.ImportFrom)) and (getattr(n, "module", None) or n.names[0].name).startswith("oracle")}
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and (getattr(n, "module", None) or n.names[0].name).startswith("oracle")]
    assert offenders == {"cpu_baseline", "cpu_baseline_unet"} and not top          # both are cpu_baseline legs


def test_oracle_unet_init_params_and_cpu_baseline_leg():
    """oracle/unet.py::init_params walks diffusers' names exactly as unet_forward consumes them — the parameter totals are the published ones
    (SD 1.5 UNet 859,520,964; SDXL UNet 2,567,463,684) — and bench.py's UNet cpu_baseline leg (BASELINE.json configs[0]) runs one full oracle
    train step and reports the contract's fields"""
    from types import SimpleNamespace

    from oracle.unet import init_params
    import bench
    assert sum(v.numel() for v in init_params(UNetConfig.sd15(), 0, shapes_only=True).values()) == 859_520_964
    shapes = init_params(UNetConfig(), 0, shapes_only=True)
    assert sum(v.numel() for v in shapes.values()) == 2_567_463_684 and shapes["add_embedding.linear_1.weight"].shape == (1280, 2816)
    out = bench.cpu_baseline_unet(SimpleNamespace(res=64, rank=4), sd15=True, lora=True)
    assert set(out) == {"value", "unit", "cores", "kind", "sample"} and out["kind"] == "port" and out["unit"] == "images/s" and out["value"] > 0
    assert "LoRA r4 on 128 attention projections" in out["sample"] and "64^2" in out["sample"]


def test_refiner_training_timestep_ranges_match_reference_code_outputs():
    """segmented_timestep_selection with refiner_training (custom_schedule.py:21-31), normal and inverted, executed by tools/gen_golden.py"""
    from pathlib import Path
    G = torch.load(Path(__file__).parent / "golden" / "ddpm_sampling_vectors.pt", weights_only=False)
    assert len(G["segmented_refiner"]) == 8
    s = DDPMSchedule()
    for invert, strength, bsz, seed, want in G["segmented_refiner"]:
        torch.manual_seed(seed)
        got = s.sample_timesteps(bsz, refiner_training=True, refiner_invert_schedule=invert, refiner_strength=strength)
        assert torch.equal(got, want), (invert, strength, bsz)
        assert (got >= int(strength * 1000)).all() if invert else (got < int(1000 * strength)).all()
    for bsz, seed, want in G["segmented"]:                                       # the base-model range is unchanged
        torch.manual_seed(seed)
        assert torch.equal(s.sample_timesteps(bsz), want)
