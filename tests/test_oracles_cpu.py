"""CPU checks of the PixArt / VAE oracle restatements (shapes, mask semantics, ControlNet wiring, posterior algebra) — the host-side logic the
GPU parity tests lean on."""
import math

import torch

from oracle import pixart as OP
from oracle import vae as OV


def _pixart_weights(cfg, n_ctrl, seed=0):
    g = torch.Generator().manual_seed(seed)
    D = cfg.D
    P = {}

    def lin(n, o, i):
        P[n + ".weight"] = torch.randn(o, i, generator=g) / math.sqrt(i)
        P[n + ".bias"] = torch.randn(o, generator=g) * 0.02

    def blk(p, Q):
        for a, kd in (("attn1.", D), ("attn2.", D)):
            for nm in ("to_q", "to_k", "to_v", "to_out.0"):
                Q[p + a + nm + ".weight"] = torch.randn(D, D, generator=g) / math.sqrt(D)
                Q[p + a + nm + ".bias"] = torch.randn(D, generator=g) * 0.02
        Q[p + "ff.net.0.proj.weight"] = torch.randn(4 * D, D, generator=g) / math.sqrt(D); Q[p + "ff.net.0.proj.bias"] = torch.zeros(4 * D)
        Q[p + "ff.net.2.weight"] = torch.randn(D, 4 * D, generator=g) / math.sqrt(4 * D); Q[p + "ff.net.2.bias"] = torch.zeros(D)
        Q[p + "scale_shift_table"] = torch.randn(6, D, generator=g) / math.sqrt(D)

    P["pos_embed.proj.weight"] = torch.randn(D, 4, 2, 2, generator=g) / 4; P["pos_embed.proj.bias"] = torch.zeros(D)
    for e, o in (("timestep_embedder", D), ("resolution_embedder", D // 3), ("aspect_ratio_embedder", D // 3)):
        lin(f"adaln_single.emb.{e}.linear_1", o, 256); lin(f"adaln_single.emb.{e}.linear_2", o, o)
    lin("adaln_single.linear", 6 * D, D); lin("caption_projection.linear_1", D, cfg.caption_channels); lin("caption_projection.linear_2", D, D)
    P["scale_shift_table"] = torch.randn(2, D, generator=g) / math.sqrt(D); lin("proj_out", 4 * cfg.out_channels, D)
    for i in range(cfg.num_layers):
        blk(f"transformer_blocks.{i}.", P)
    C = {}
    for i in range(n_ctrl):
        p = f"controlnet_blocks.{i}."
        blk(p + "transformer_block.", C)
        if i == 0:
            C[p + "before_proj.weight"], C[p + "before_proj.bias"] = torch.zeros(D, D), torch.zeros(D)
        C[p + "after_proj.weight"], C[p + "after_proj.bias"] = torch.zeros(D, D), torch.zeros(D)
    return P, C


def test_pixart_oracle_mask_and_zero_init_controlnet():
    cfg = OP.PixArtConfig(num_attention_heads=2, attention_head_dim=24, num_layers=3, caption_channels=32, cross_attention_dim=48, sample_size=128)
    P, C = _pixart_weights(cfg, 2)
    g = torch.Generator().manual_seed(1)
    lat, cond = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    enc = torch.randn(2, 6, 32, generator=g)
    mask = torch.tensor([[1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 0]], dtype=torch.float32)
    t = torch.tensor([10.0, 700.0])
    res, ar = torch.tensor([[8.0, 8.0]]).expand(2, -1), torch.tensor([[1.0]]).expand(2, -1)
    base = OP.pixart_forward(P, cfg, lat, enc, mask, t, res, ar)
    assert base.shape == (2, 8, 8, 8)
    # masked text tokens must not influence the output (bias -10000 -> exp underflows to exactly 0 in fp32)
    enc2 = enc.clone(); enc2[0, 3:] += 5.0; enc2[1, 5:] -= 3.0
    assert torch.allclose(base, OP.pixart_forward(P, cfg, lat, enc2, mask, t, res, ar), atol=1e-5)
    # zero-initialised before/after projections: the ControlNet wrapper reproduces the trunk exactly (pixart/controlnet.py:37-40, 58-60)
    ctl = OP.controlnet_forward(P, C, cfg, 2, lat, cond, enc, mask, t, res, ar)
    assert torch.allclose(base, ctl, atol=1e-5)
    C["controlnet_blocks.0.after_proj.weight"] += 0.05 * torch.randn(cfg.D, cfg.D, generator=g)
    assert not torch.allclose(base, OP.controlnet_forward(P, C, cfg, 2, lat, cond, enc, mask, t, res, ar), atol=1e-4)
    assert OP.pixart_flops_fwd(OP.PixArtConfig(sample_size=256), 256, 256) / 1e12 > 50       # F_fwd ≈ 5.2e13 at 2K


def test_vae_posterior_and_scaling():
    m = torch.randn(2, 8, 4, 4)
    m[:, 4:] = torch.tensor(40.0)                                     # logvar above the clamp
    eps = torch.ones(2, 4, 4, 4)
    z = OV.sample_and_scale(m, OV.VAEConfig(), eps)
    assert torch.allclose(z, (m[:, :4] + math.exp(10.0)) * 0.13025)   # clamp(logvar, max 20) -> std = e^10
    zf = OV.sample_and_scale(m, OV.VAEConfig.flux(), None)
    assert torch.allclose(zf, (m[:, :4] - 0.1159) * 0.3611)           # mode + (z - shift) * scale (foundation_mixins.py:72-74)
    assert abs(OV.encoder_flops(OV.VAEConfig(), 1024, 1024) / 1e12 - 4.9) < 0.3


def test_vae_oracle_parameter_totals_and_decoder_roundtrip_shapes():
    """oracle/vae.py::init_params walks diffusers' AutoencoderKL names (encoder + decoder + quant convs): the totals are the published sizes
    (SD / SDXL VAE 83,653,863; FLUX.1 VAE 83,819,683), and encode -> scale -> unscale -> decode closes on the pixel grid"""
    import torch

    from oracle.vae import VAEConfig, decode, encode_moments, init_params, sample_and_scale, unscale_latents
    assert sum(v.numel() for v in init_params(VAEConfig(), 0, shapes_only=True).values()) == 83_653_863
    assert sum(v.numel() for v in init_params(VAEConfig.flux(), 0, shapes_only=True).values()) == 83_819_683
    for cfg in (VAEConfig(block_out_channels=(32, 64, 64, 64)), VAEConfig(latent_channels=16, scaling_factor=0.3611, shift_factor=0.1159, use_quant_conv=False,
                                                                           block_out_channels=(32, 64, 64, 64))):
        P = init_params(cfg, 1)
        assert ("post_quant_conv.weight" in P) == cfg.use_quant_conv
        x = torch.randn(2, 3, 64, 48, generator=torch.Generator().manual_seed(0)).clamp(-1, 1)
        z = sample_and_scale(encode_moments(P, cfg, x), cfg)
        zr = unscale_latents(z, cfg)
        mean = encode_moments(P, cfg, x).chunk(2, dim=1)[0]
        torch.testing.assert_close(zr, mean, rtol=1e-5, atol=1e-5)               # unscale inverts scale_vae_latents_for_cache
        y = decode(P, cfg, zr)
        assert z.shape == (2, cfg.latent_channels, 8, 6) and y.shape == x.shape and torch.isfinite(y).all()


def test_oracle_architectures_have_the_published_parameter_totals():
    """The network restatements cannot be pinned to reference tensors (diffusers is un-vendored), but their parameter name / shape walks can be
    pinned to public facts: the exact parameter totals of the released checkpoints.  FLUX.1-dev transformer 11,901,408,320; SD3-Medium MMDiT
    2,028,328,000 (UNet and VAE totals: tests/test_unet_cpu.py, test above)."""
    import math

    from oracle import flux as OF
    from oracle import sd3 as OS
    assert sum(math.prod(s) for s in OF.param_shapes(OF.FluxConfig()).values()) == 11_901_408_320
    sd3m = OS.SD3Config(sample_size=128, num_layers=24, attention_head_dim=64, num_attention_heads=24, joint_attention_dim=4096,
                        pooled_projection_dim=2048, pos_embed_max_size=192)
    assert sum(math.prod(s) for s in OS.param_shapes(sd3m).values()) == 2_028_328_000


def test_pixart_oracle_names_and_published_total():
    """PixArt-Sigma-XL-2: 610,856,096 parameters (no additional size conditions); with them (the 1024 alpha-style config) 611,349,152; the name
    walk equals what pixart_forward reads (a forward over param_shapes-built weights runs)"""
    import math

    import torch

    from oracle.pixart import PixArtConfig, param_shapes, pixart_forward
    assert sum(math.prod(s) for s in param_shapes(PixArtConfig(use_additional_conditions=False)).values()) == 610_856_096
    assert sum(math.prod(s) for s in param_shapes(PixArtConfig(use_additional_conditions=True)).values()) == 611_349_152
    cfg = PixArtConfig(num_attention_heads=2, attention_head_dim=24, num_layers=2, cross_attention_dim=48, sample_size=16, caption_channels=32,
                       use_additional_conditions=True)
    g = torch.Generator().manual_seed(0)
    P = {k: torch.randn(*s, generator=g) * 0.05 for k, s in param_shapes(cfg).items()}
    out = pixart_forward(P, cfg, torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 5, 32, generator=g), torch.ones(2, 5), torch.tensor([500.0]),
                         resolution=torch.tensor([[64.0, 64.0]] * 2), aspect_ratio=torch.tensor([[1.0]] * 2))
    assert out.shape == (2, 8, 8, 8) and torch.isfinite(out).all()
