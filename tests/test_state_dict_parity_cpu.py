"""Checkpoint surface of the PRODUCT models, checked without a GPU (parameters are plain torch tensors; only forward / backward need the device):
their state-dict names and shapes equal the oracle's walks of diffusers' layouts, and their totals equal the published sizes of the released
checkpoints — FLUX.1-dev 11,901,408,320; SD3-Medium 2,028,328,000; PixArt-Sigma (1024 config, with size conditions) 611,349,152; SD 1.5 UNet
859,520,964 — so a released checkpoint's keys land one-to-one.  For the UNet the native-layout converter (conv [O,I,3,3] <-> [O, 9*I], padded
conv_in / conv_out, K-major copies) is additionally round-tripped bit-exactly at full SD 1.5 size."""
import math

import torch

from oracle import flux as OF
from oracle import pixart as OP
from oracle import sd3 as OS
from oracle.unet import UNetConfig, init_params


def _named(m):
    return {n: tuple(p.shape) for n, p in m.named_parameters()}


def test_transformer_families_match_oracle_walks_and_published_totals():
    from simpletuner_amd.flux.transformer import FluxTransformer2DModel
    from simpletuner_amd.pixart.transformer import PixArtTransformer2DModel
    from simpletuner_amd.sd3.transformer import SD3Transformer2DModel
    flux = _named(FluxTransformer2DModel(device="meta", guidance_embeds=True))
    assert flux == {k: tuple(v) for k, v in OF.param_shapes(OF.FluxConfig()).items()} and sum(math.prod(s) for s in flux.values()) == 11_901_408_320
    sd3 = _named(SD3Transformer2DModel(device="meta", sample_size=128, num_layers=24, num_attention_heads=24, attention_head_dim=64,
                                       caption_projection_dim=1536, pooled_projection_dim=2048, pos_embed_max_size=192))
    ref = OS.param_shapes(OS.SD3Config(sample_size=128, num_layers=24, attention_head_dim=64, num_attention_heads=24, joint_attention_dim=4096,
                                       pooled_projection_dim=2048, pos_embed_max_size=192))
    assert sd3 == {k: tuple(v) for k, v in ref.items()} and sum(math.prod(s) for s in sd3.values()) == 2_028_328_000
    pix = _named(PixArtTransformer2DModel(device="meta", sample_size=128))
    assert pix == {k: tuple(v) for k, v in OP.param_shapes(OP.PixArtConfig(sample_size=128)).items()}
    assert sum(math.prod(s) for s in pix.values()) == 611_349_152


def test_sd15_unet_state_dict_surface_and_converter_roundtrip():
    from simpletuner_amd.sd1x.model import SD15_ARCH
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    m = UNet2DConditionModel(device="cpu", **SD15_ARCH)
    sd = m.diffusers_state_dict()
    ref = init_params(UNetConfig.sd15(), 0, shapes_only=True)
    assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
    assert sum(v.numel() for v in sd.values()) == 859_520_964
    g = torch.Generator().manual_seed(0)
    new = {k: (torch.randn(v.shape, generator=g) * 0.02).to(torch.bfloat16) for k, v in sd.items()}
    m.load_diffusers_state(new)
    back = m.diffusers_state_dict()
    assert all(torch.equal(back[k].to(torch.bfloat16), new[k]) for k in new)


def test_small_sdxl_style_unet_surface():
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    small = dict(block_out_channels=(64, 128), layers_per_block=1, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
                 up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2), attention_head_dim=(1, 2), cross_attention_dim=128,
                 projection_class_embeddings_input_dim=64 + 6 * 64, addition_time_embed_dim=64)
    sd = UNet2DConditionModel(device="cpu", **small).diffusers_state_dict()
    ref = init_params(UNetConfig(**small), 0, shapes_only=True)
    assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)


def test_vae_encoder_state_dict_surface():
    """the product AutoencoderKL reads exactly diffusers' encoder (+ quant_conv) keys, with their shapes, for the SDXL and FLUX.1 layouts"""
    from oracle.vae import VAEConfig
    from oracle.vae import init_params as vae_params
    from simpletuner_amd.vae.autoencoder_kl import AutoencoderKL
    for cfg, kw in ((VAEConfig(), {}), (VAEConfig.flux(), dict(latent_channels=16, scaling_factor=0.3611, shift_factor=0.1159, use_quant_conv=False))):
        sd = AutoencoderKL(device="cpu", **kw).synthetic_state_dict(0)
        ref = {k: v for k, v in vae_params(cfg, 0, shapes_only=True).items() if k.startswith(("encoder.", "quant_conv"))}
        assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)


def test_lora_adapter_surfaces_use_peft_names_on_the_reference_target_sets():
    """add_lora_adapter (common.py:1049-1128): peft key names `<module>.lora_A.default.weight` / `.lora_B.default.weight` on exactly the target
    modules — Flux 'default' (to_q/k/v, to_out.0; single blocks to_q/k/v) and 'all' (+ add_*_proj, to_add_out) = oracle.flux.lora_targets; at
    FLUX.1-dev size that is 19*4 + 38*3 = 190 wrapped Linears.  (The UNet's attach builds K-major copies on the
    device: its 128-module surface is checked by tests/test_unet_model_gpu.py.)"""
    from simpletuner_amd.flux.transformer import FluxTransformer2DModel
    from tests import parity_utils as PU
    for which in ("default", "all"):
        m = FluxTransformer2DModel(device="cpu", **PU.small_flux_cfg(layers=2, single=3))
        params = m.add_lora_adapter(rank=8, alpha=8.0, targets=which, seed=1)
        names = [n for n, _ in m.named_parameters() if ".lora_" in n]
        mods = {n.split(".lora_")[0] for n in names}
        assert mods == set(OF.lora_targets(PU.oracle_cfg(m), which)) and len(params) == len(names) == 2 * len(mods)
        assert all(n.endswith((".lora_A.default.weight", ".lora_B.default.weight")) for n in names)
        a = dict(m.named_parameters())["transformer_blocks.0.attn.to_q.lora_A.default.weight"]
        b = dict(m.named_parameters())["transformer_blocks.0.attn.to_q.lora_B.default.weight"]
        assert a.shape == (8, 256) and b.shape == (256, 8) and a.requires_grad and b.requires_grad
    assert len(OF.lora_targets(OF.FluxConfig(), "default")) == 190


def test_transformer_checkpoint_loaders_fill_fused_storage_through_diffusers_keys():
    """load_flat_state(diffusers-keyed state dict): every per-projection parameter (a VIEW into fused qkv / modulation / arena storage) ends up
    holding exactly the bf16 value of its checkpoint tensor — Flux, SD3 and PixArt, random weights in the oracle's (= diffusers') names"""
    from simpletuner_amd.flux.transformer import FluxTransformer2DModel
    from simpletuner_amd.pixart.transformer import PixArtTransformer2DModel
    from simpletuner_amd.sd3.transformer import SD3Transformer2DModel
    from tests import parity_utils as PU

    def check(model, state):
        model.load_flat_state(state)
        bad = [n for n, p in model.named_parameters() if not torch.equal(p.detach().float(), state[n].to(p.dtype).float())]
        assert not bad, bad[:3]

    g = torch.Generator().manual_seed(3)
    rnd = lambda shapes: {k: torch.randn(*s, generator=g) * 0.05 for k, s in shapes.items()}
    fl = FluxTransformer2DModel(device="cpu", **PU.small_flux_cfg(layers=2, single=2))
    check(fl, rnd(OF.param_shapes(PU.oracle_cfg(fl))))
    kw = dict(sample_size=32, num_layers=3, num_attention_heads=2, attention_head_dim=64, joint_attention_dim=128, pooled_projection_dim=64, pos_embed_max_size=24)
    sd3 = SD3Transformer2DModel(device="cpu", caption_projection_dim=128, **kw)
    st = rnd(OS.param_shapes(OS.SD3Config(**kw)))
    if hasattr(sd3, "pos_embed") and hasattr(sd3.pos_embed, "pos_embed"):
        st["pos_embed.pos_embed"] = sd3.pos_embed.pos_embed.detach().float().cpu().clone()          # a buffer of the checkpoint, not a parameter
    check(sd3, st)
    pkw = dict(num_attention_heads=8, attention_head_dim=72, num_layers=2, cross_attention_dim=576, sample_size=16, caption_channels=64, use_additional_conditions=True)
    pcfg = OP.PixArtConfig(**pkw)
    pix = PixArtTransformer2DModel(device="cpu", **pkw)
    check(pix, rnd(OP.param_shapes(pcfg)))


def test_sdxl_unet_state_dict_surface_at_full_size():
    """SDXL UNet (BASELINE.json configs[1]): the product's diffusers-keyed surface at full size — 1,680 tensors, 2,567,463,684 parameters"""
    import psutil
    import pytest
    if psutil.virtual_memory().available < 24 << 30:
        pytest.skip("needs ~11 GB of host memory")
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    sd = UNet2DConditionModel(device="cpu").diffusers_state_dict()
    ref = init_params(UNetConfig(), 0, shapes_only=True)
    assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
    assert sum(v.numel() for v in sd.values()) == 2_567_463_684


def test_pixart_controlnet_adapter_surface_and_init():
    """PixArtSigmaControlNetTransformerModel (pixart/controlnet.py:17-163): adapter keys `controlnet.controlnet_blocks.{i}.before_proj (block 0 only) |
    transformer_block.* | after_proj`, zero-initialised projections, blocks copied from trunk blocks 0..N-1 (`from_transformer`), trunk frozen"""
    from simpletuner_amd.pixart.transformer import PixArtSigmaControlNetTransformerModel, PixArtTransformer2DModel
    pkw = dict(num_attention_heads=8, attention_head_dim=72, num_layers=4, cross_attention_dim=576, sample_size=16, caption_channels=64, use_additional_conditions=True)
    trunk = PixArtTransformer2DModel(device="cpu", **pkw)
    g = torch.Generator().manual_seed(1)
    trunk.load_flat_state({k: torch.randn(*s, generator=g) * 0.05 for k, s in OP.param_shapes(OP.PixArtConfig(**pkw)).items()})
    wrap = PixArtSigmaControlNetTransformerModel(trunk, num_layers=2, init_from_transformer=True)
    named = dict(wrap.named_parameters())
    train = {n for n, p in named.items() if p.requires_grad}
    assert train and all(n.startswith("controlnet.controlnet_blocks.") for n in train)
    assert not [n for n, p in named.items() if not n.startswith("controlnet.") and p.requires_grad]                      # trunk frozen
    assert "controlnet.controlnet_blocks.0.before_proj.weight" in train and "controlnet.controlnet_blocks.1.before_proj.weight" not in named
    for i in range(2):
        for leaf in ("after_proj.weight", "after_proj.bias") + (("before_proj.weight", "before_proj.bias") if i == 0 else ()):
            assert torch.count_nonzero(named[f"controlnet.controlnet_blocks.{i}.{leaf}"]) == 0
        pre = f"controlnet.controlnet_blocks.{i}.transformer_block."
        for n in [n for n in train if n.startswith(pre)]:
            src = dict(trunk.named_parameters())[f"transformer_blocks.{i}." + n[len(pre):]]
            assert torch.equal(named[n].detach().float(), src.detach().float()), n
    per_block = {n[len("controlnet.controlnet_blocks.1."):] for n in train if n.startswith("controlnet.controlnet_blocks.1.")}
    want = {"transformer_block." + k[len("transformer_blocks.0."):] for k in OP.param_shapes(OP.PixArtConfig(**pkw)) if k.startswith("transformer_blocks.0.")}
    assert per_block == want | {"after_proj.weight", "after_proj.bias"}
