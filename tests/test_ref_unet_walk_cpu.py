"""oracle/unet.py's DOWN HALF of the UNet2DConditionModel walk against reference code executed in the build container.

The reference never vendors diffusers' UNet2DConditionModel, but it vendors diffusers' ControlNetModel (simpletuner/helpers/models/kolors/controlnet.py:132-931), whose
constructor and forward are the UNet's up to the mid block.  tools/gen_ref_unet_walk.py imports that file unmodified and runs it on a small SDXL-shaped configuration
(DownBlock2D + CrossAttnDownBlock2D, "text_time" addition embedding, linear projections) with its 1x1 output convolutions set to the identity; the down / mid blocks it
asks `get_down_block` for are stand-ins that compose this oracle's own leaves (pinned one by one in tests/test_ref_unet_leaves_cpu.py) and assert the constructor's
arguments.  Pinned here, forward and every gradient <= 1e-5: timestep projection -> embedder, `time_ids.flatten() -> add_time_proj -> reshape -> cat([text_embeds, .]) ->
add_embedding`, emb = time + aug, conv_in, the skip order `(sample,) + res_samples` of every block, the mid block, and the head-count meaning of `attention_head_dim`.
Not pinned by this: the order of operations INSIDE a down block and the up path (restated from the published diffusers modules)."""
import os

import pytest
import torch

from oracle import unet as OU
from tests.ref_fixture_utils import rel_l2

TOL = 1e-5
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_unet_walk.pt")


def _load():
    G = torch.load(GOLD, weights_only=False)
    c = dict(G["config"])
    for k in ("block_out_channels", "down_block_types", "up_block_types", "transformer_layers_per_block", "attention_head_dim"):
        c[k] = tuple(c[k])
    return G, OU.UNetConfig(**c)


def test_down_half_of_the_unet_walk_matches_the_executed_reference_controlnet():
    G, cfg = _load()
    P = {k: v.clone().requires_grad_(True) for k, v in G["params"].items()}
    I = G["inputs"]
    leaves = {k: I[k].clone().requires_grad_(True) for k in ("sample", "encoder_hidden_states", "text_embeds")}
    skips, mid, _ = OU.unet_down_mid(P, cfg, leaves["sample"], I["timestep"], leaves["encoder_hidden_states"],
                                     {"text_embeds": leaves["text_embeds"], "time_ids": I["time_ids"]})
    assert len(skips) == len(G["down"]) == 1 + sum(cfg.layers_per_block + (1 if i < len(cfg.block_out_channels) - 1 else 0) for i in range(len(cfg.block_out_channels)))
    worst = 0.0
    for i, (a, b) in enumerate(zip(skips, G["down"])):
        assert a.shape == b.shape, i
        worst = max(worst, rel_l2(a, b))
        assert rel_l2(a, b) <= TOL, (i, rel_l2(a, b))
    assert rel_l2(mid, G["mid"]) <= TOL
    sum((t * w).sum() for t, w in zip(list(skips) + [mid], G["w"])).backward()
    gw = (0.0, "")
    for k, g in G["grads"].items():
        r = rel_l2(P[k].grad, g)
        gw = max(gw, (r, k))
        assert r <= 5 * TOL, (k, r)
    for k, g in G["input_grads"].items():
        assert rel_l2(leaves[k].grad, g) <= 5 * TOL, k
    # the up half consumes exactly these tensors: unet_forward runs through unet_down_mid (one code path)
    with torch.no_grad():
        full = OU.init_params(cfg, seed=5)
        full.update({k: v.detach() for k, v in P.items()})
        out = OU.unet_forward(full, cfg, I["sample"], I["timestep"], I["encoder_hidden_states"], {"text_embeds": I["text_embeds"], "time_ids": I["time_ids"]})
    assert out.shape == I["sample"].shape and torch.isfinite(out).all()
    print(f"[pinned] UNet down walk: {len(skips)} skip tensors + mid, worst rel-L2 {worst:.2e}; {len(G['grads'])} parameter gradients, worst {gw[0]:.2e} ({gw[1]})")


def test_the_reference_constructor_registered_the_configuration_the_oracle_was_given():
    """what ControlNetModel.__init__ received (recorded by @register_to_config): the keys the oracle's UNetConfig mirrors, with the same values"""
    G, cfg = _load()
    rc = G["registered_config"]
    assert tuple(rc["block_out_channels"]) == cfg.block_out_channels and tuple(rc["down_block_types"]) == cfg.down_block_types
    assert rc["layers_per_block"] == cfg.layers_per_block and rc["cross_attention_dim"] == cfg.cross_attention_dim
    # the head count rides in `attention_head_dim`: the constructor re-registers it as num_attention_heads (controlnet.py:251, :293)
    assert tuple(rc["attention_head_dim"]) == cfg.attention_head_dim == tuple(rc["num_attention_heads"])
    assert tuple(rc["transformer_layers_per_block"]) == cfg.transformer_layers_per_block and rc["use_linear_projection"] == cfg.use_linear_projection
    assert rc["addition_embed_type"] == "text_time" and rc["addition_time_embed_dim"] == cfg.addition_time_embed_dim
    assert rc["projection_class_embeddings_input_dim"] == cfg.projection_class_embeddings_input_dim == G["inputs"]["text_embeds"].shape[1] + 6 * cfg.addition_time_embed_dim
    assert rc["norm_num_groups"] == cfg.norm_num_groups and rc["norm_eps"] == cfg.norm_eps and rc["flip_sin_to_cos"] is True and rc["freq_shift"] == 0


def test_a_wrong_skip_order_or_embedding_order_is_detected():
    """the pin has teeth: swapping the concatenation order of the text_time embedding, or treating attention_head_dim as the head WIDTH, moves the outputs"""
    G, cfg = _load()
    P, I = G["params"], G["inputs"]
    with torch.no_grad():
        ref_mid = G["mid"]
        # (a) head width instead of head count: 32 channels / width 4 = 8 heads instead of 4
        bad = OU.UNetConfig(**{**cfg.__dict__, "attention_head_dim": (2, 8)})
        _, mid_b, _ = OU.unet_down_mid(P, bad, I["sample"], I["timestep"], I["encoder_hidden_states"], {"text_embeds": I["text_embeds"], "time_ids": I["time_ids"]})
        assert rel_l2(mid_b, ref_mid) > 1e-3
        # (b) time ids in another order
        _, mid_c, _ = OU.unet_down_mid(P, cfg, I["sample"], I["timestep"], I["encoder_hidden_states"], {"text_embeds": I["text_embeds"], "time_ids": I["time_ids"].flip(1)})
        assert rel_l2(mid_c, ref_mid) > 1e-3
