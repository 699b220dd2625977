"""AutoencoderKL.encode on the HIP path vs the fp32 oracle restatement (oracle/vae.py) on identical weights / pixels.
PARITY UNPINNED against the reference (diffusers un-vendored; no golden tensors): bf16 HIP vs fp32 oracle, moments rel-L2 <= 2e-2 (stated here)."""
import pytest
import torch

from oracle.vae import VAEConfig, encode_moments, sample_and_scale

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("kind", ["sdxl", "flux"])
def test_vae_encode_matches_oracle(kind):
    from simpletuner_amd.vae.autoencoder_kl import AutoencoderKL
    dev = "cuda:0"
    cfg = VAEConfig(block_out_channels=(64, 128, 128, 128)) if kind == "sdxl" else VAEConfig(latent_channels=16, scaling_factor=0.3611, shift_factor=0.1159,
                                                                                               use_quant_conv=False, block_out_channels=(64, 128, 128, 128))
    vae = AutoencoderKL(latent_channels=cfg.latent_channels, block_out_channels=cfg.block_out_channels, scaling_factor=cfg.scaling_factor,
                        shift_factor=cfg.shift_factor, use_quant_conv=cfg.use_quant_conv, device=dev)
    sd = vae.synthetic_state_dict(3)
    vae.load_state_dict(sd)
    x = torch.randn(2, 3, 96, 64, generator=torch.Generator().manual_seed(1)).clamp(-1, 1)
    got = vae.encode_moments(x.to(dev))
    ref = encode_moments(sd, cfg, x.to(torch.bfloat16).float())
    assert got.shape == ref.shape == (2, 2 * cfg.latent_channels, 12, 8)
    r = _rel(got.cpu(), ref)
    print(f"[vae {kind}] moments rel-L2 {r:.3e}")
    assert r < 2e-2
    # the reference's seam: .latent_dist.mode() / scaling
    z = vae.encode_scaled(x.to(dev), sample=False)
    zr = sample_and_scale(ref, cfg)
    assert _rel(z.cpu(), zr) < 2.5e-2
    d = vae.encode(x.to(dev)).latent_dist
    s = d.sample(generator=torch.Generator(device=dev).manual_seed(0))
    assert s.shape == (2, cfg.latent_channels, 12, 8) and torch.isfinite(s.float()).all()


def test_softmax_rows_kernel():
    from simpletuner_amd import ops
    dev = "cuda:0"
    for n in (8, 96, 2048, 4104, 16384):
        x = (torch.randn(37, n, device=dev) * 3).to(torch.bfloat16)
        ref = torch.softmax(x.float() * 0.7, dim=-1)
        got = ops.softmax_rows_(x.clone(), 0.7)
        assert (got.float() - ref).abs().max().item() < 4e-3 and abs(got.float().sum(-1).mean().item() - 1.0) < 2e-3


def test_plugin_vae_seam_encodes_and_scales_like_encode_scaled():
    """ModelFoundation.load_vae / encode_with_vae / scale_vae_latents_for_cache (common.py:2663-2772, foundation_mixins.py:66-79) on the Flux layout:
    the three-call form of the reference's VAECache equals the fused encode_scaled() pass"""
    from simpletuner_amd.flux.model import Flux
    from simpletuner_amd.training.trainer import St355Accelerator, default_config
    dev = torch.device("cuda", 0)
    pl = Flux(default_config(model_family="flux"), St355Accelerator(dev))
    vae = pl.get_vae()
    assert vae is pl.vae and vae.config.latent_channels == 16 and vae.config.shift_factor == 0.1159
    x = (torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(torch.bfloat16).to(dev)
    dist = pl.encode_with_vae(vae, x).latent_dist
    z = pl.scale_vae_latents_for_cache(dist.mode(), vae)
    assert z.shape == (1, 16, 8, 8) and torch.isfinite(z.float()).all()
    want = vae.encode_scaled(x, sample=False)
    assert torch.allclose(z.float(), want.float(), rtol=2e-2, atol=2e-2)
