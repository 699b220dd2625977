"""oracle/unet.py's leaves against reference code executed in the build container (tools/gen_ref_unet_leaves.py -> tests/golden/ref_unet_leaves.pt).

Reference code in the fixture: ResnetBlock / Upsample / Downsample / AttnBlock of the vendored KL autoencoder (helpers/models/ideogram/autoencoder.py:29-110)
fed with diffusers-named tensors through its own convert_diffusers_state_dict (:321-392); Timesteps / TimestepEmbedding lifted from
helpers/models/heartmula/codec/transformer.py:15-25, 410-440.  The `basic_block_shim` case is tools/ref_shim.py's independent restatement of diffusers'
BasicTransformerBlock (layer_norm / GEGLU / cross-attention) — a guard against transcription slips, not reference code (the oracle header says so).
Tolerance: fp32 vs fp32, rel-L2 <= 1e-5 on the output and on every gradient."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import unet as OU
from oracle.flux import timestep_proj
from tests.ref_fixture_utils import rel_l2

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_unet_leaves.pt")
TOL = 1e-5


@pytest.fixture(scope="module")
def cases():
    return torch.load(GOLD, weights_only=False)["cases"]


def _check(case, fn):
    leaves = {k: v.clone().requires_grad_(True) for k, v in {**case["inputs"], **case["params"]}.items()}
    out = fn(leaves)
    assert rel_l2(out, case["out"]) <= TOL, rel_l2(out, case["out"])
    (out * case["w"]).sum().backward()
    for k, g in case["grads"].items():
        got = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        if float(g.double().norm()) <= 1e-5 * g.numel() ** 0.5:
            # analytically-zero gradients (a conv bias in front of a one-channel-per-group GroupNorm is removed by the mean): both sides hold rounding noise only
            assert float(got.double().norm()) <= 1e-4 * g.numel() ** 0.5, k
        else:
            assert rel_l2(got, g) <= TOL, (k, rel_l2(got, g))


@pytest.mark.parametrize("tag", ["resnet_same", "resnet_widen"])
def test_resnet_is_the_vendored_resnet_block(cases, tag):
    c = cases[tag]
    pre = "encoder.mid_block.resnets.0."
    co = c["params"][pre + "conv1.weight"].shape[0]

    def fn(L):
        P = dict(L)
        emb = torch.zeros(L["x"].shape[0], 16)
        P[pre + "time_emb_proj.weight"] = torch.zeros(co, 16)          # the vendored block has no time-embedding term
        P[pre + "time_emb_proj.bias"] = torch.zeros(co)
        return OU.resnet(P, pre, L["x"], emb, 32, 1e-6)
    _check(c, fn)


def test_upsample_is_the_vendored_upsample(cases):
    _check(cases["upsample"], lambda L: OU.upsample(L, "decoder.up_blocks.0.upsamplers.0.conv", L["x"]))


def test_downsample_convolution_is_the_vendored_one_up_to_the_padding_choice(cases):
    """the vendored (VAE) form pads (0,1,0,1) and convolves with padding 0; the UNet's Downsample2D convolves with padding 1 — the stride-2 3x3 convolution is
    the pinned part, `padding=1` in oracle.unet.downsample is restated"""
    _check(cases["downsample_vae_padding"], lambda L: OU.downsample(L, "encoder.down_blocks.0.downsamplers.0.conv", F.pad(L["x"], (0, 1, 0, 1)), padding=0))


def test_group_norm_attention_residual_is_the_vendored_attn_block(cases):
    pre = "encoder.mid_block.attentions.0."

    def fn(L):
        x = L["x"]
        B, C, H, W = x.shape
        n = F.group_norm(x, 32, L[pre + "group_norm.weight"], L[pre + "group_norm.bias"], 1e-6).permute(0, 2, 3, 1).reshape(B, H * W, C)
        return x + OU._attention(L, pre, n, n, 1).reshape(B, H, W, C).permute(0, 3, 1, 2)
    _check(cases["attn_block"], fn)


def test_timestep_embedding_is_the_lifted_one(cases):
    c = cases["timestep_embedding"]
    t = c["inputs"]["t"]
    assert rel_l2(timestep_proj(t, c["sinusoid"].shape[1]), c["sinusoid"]) <= TOL
    _check(c, lambda L: OU.time_embedding(L, "time_embedding", timestep_proj(L["t"], c["sinusoid"].shape[1])))


def test_basic_block_agrees_with_the_independent_shim_restatement(cases):
    c = cases["basic_block_shim"]
    _check(c, lambda L: OU.basic_block(L, "transformer_blocks.0.", L["h"], L["ctx"], c["heads"]))
