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
