"""AutoencoderKL.decode's HOST SEQUENCING on the CPU (see tests/ops_emulator.py): post_quant_conv, conv_in, the mid block with its single-head attention, four up blocks with
nearest-2x upsampling, conv_out — the validation path's last step — against the oracle's decoder, same weights / latents.  (The encoder is ONE C entry
point, st355_vae_encode: its sequencing lives in libst355 and is proven on the GPU against the per-kernel sequencing, tests/test_vae_gpu.py.)"""
import pytest
import torch

from oracle.vae import VAEConfig, decode, unscale_latents
from tests import ops_emulator as EMU


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("kind", ["sdxl", "flux"])
def test_decode_through_the_emulator_matches_the_oracle(monkeypatch, kind):
    EMU.install(monkeypatch)
    from simpletuner_amd.vae.autoencoder_kl import AutoencoderKL
    cfg = VAEConfig(block_out_channels=(64, 128, 128, 128)) if kind == "sdxl" else VAEConfig(latent_channels=16, scaling_factor=0.3611, shift_factor=0.1159,
                                                                                               use_quant_conv=False, block_out_channels=(64, 128, 128, 128))
    vae = AutoencoderKL(latent_channels=cfg.latent_channels, block_out_channels=cfg.block_out_channels, scaling_factor=cfg.scaling_factor,
                        shift_factor=cfg.shift_factor, use_quant_conv=cfg.use_quant_conv, device="cpu")
    sd = vae.synthetic_state_dict(5, decoder=True)
    vae.load_state_dict(sd)
    z = torch.randn(2, cfg.latent_channels, 12, 8, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16)
    got = vae.decode(z).sample
    ref = decode(sd, cfg, z.float())
    assert got.shape == ref.shape == (2, 3, 96, 64) and _rel(got, ref) < 2e-2
    zs = (torch.randn(1, cfg.latent_channels, 8, 8, generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.bfloat16)
    assert _rel(vae.decode_scaled(zs), decode(sd, cfg, unscale_latents(zs.float(), cfg).to(torch.bfloat16).float())) < 2.5e-2
