"""CPU checks of the SD3 oracle (test infrastructure) against the pieces of the reference that pin anything for this path:
the unpatchify einsum (sd3/transformer.py:879-902), the chunk orders of the two AdaLN flavours (sd3/transformer.py:126-142), the
context_pre_only last block (:174-176, :214-215), and the host-side sincos table of the device model vs the oracle's numpy form."""
import torch

from oracle import sd3 as OS


def _cfg(layers=2):
    return OS.SD3Config(sample_size=32, num_layers=layers, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=128,
                        pooled_projection_dim=64, pos_embed_max_size=24)


def test_forward_shapes_and_last_block_has_no_context_output():
    cfg = _cfg(2)
    P = OS.init_params(cfg, seed=1)
    assert "transformer_blocks.1.attn.to_add_out.weight" not in P and "transformer_blocks.1.ff_context.net.2.weight" not in P
    assert P["transformer_blocks.1.norm1_context.linear.weight"].shape[0] == 2 * cfg.inner_dim
    assert P["transformer_blocks.0.norm1_context.linear.weight"].shape[0] == 6 * cfg.inner_dim
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(2, 16, 16, 24, generator=g); enc = torch.randn(2, 20, 128, generator=g); pooled = torch.randn(2, 64, generator=g)
    t = torch.tensor([100.0, 900.0])
    out = OS.sd3_forward(P, cfg, lat, enc, pooled, t)
    assert out.shape == lat.shape and torch.isfinite(out).all()
    # context tokens only matter through attention: perturbing them changes the output, and the last block returns enc=None
    enc2 = enc.clone(); enc2[:, 0] += 1.0
    assert (OS.sd3_forward(P, cfg, lat, enc2, pooled, t) - out).abs().max() > 1e-6
    e, h = OS.joint_block(P, cfg, 1, torch.randn(2, 96, 128), torch.randn(2, 20, 128), torch.randn(2, 128))
    assert e is None and h.shape == (2, 96, 128)


def test_unpatchify_matches_reference_einsum():
    B, h, w, p, C = 2, 3, 5, 2, 16
    x = torch.randn(B, h * w, p * p * C)
    ref = torch.einsum("nhwpqc->nchpwq", x.reshape(B, h, w, p, p, C)).reshape(B, C, h * p, w * p)     # verbatim from the reference
    for b, y, xx, c in [(0, 1, 2, 3), (1, 5, 9, 15)]:
        hh, ph, ww, pw = y // 2, y % 2, xx // 2, xx % 2
        assert ref[b, c, y, xx] == x[b, hh * w + ww, (ph * 2 + pw) * C + c]


def test_sincos_table_host_and_oracle_agree():
    from simpletuner_amd.sd3.transformer import sincos_2d
    a = sincos_2d(128, 24, 16)
    b = OS.sincos_2d(128, 24, 16)
    assert torch.allclose(a, b, atol=1e-6)
    crop = OS.cropped_pos_embed(b, 24, 8, 12)
    assert torch.equal(crop[0], b.view(24, 24, -1)[8, 6]) and crop.shape == (96, 128)
