#!/usr/bin/env python3
"""Generate tests/golden/ref_unet_leaves.pt: the LEAVES of oracle/unet.py against reference code executed in this container.

    python tools/gen_ref_unet_leaves.py

The conv UNet itself (diffusers' UNet2DConditionModel) is imported, not vendored, by the reference (sdxl/model.py:306-373, sd1x/model.py:224-270,
unet_flowmap.py:23-44), so it cannot be executed here.  Its leaves can be cross-checked against code the reference DOES carry:
  * ResnetBlock / Upsample / Downsample / AttnBlock of the KL autoencoder the reference vendors (helpers/models/ideogram/autoencoder.py:29-110), fed with
    diffusers-named tensors through its own `convert_diffusers_state_dict` (:321-392) — the reference's statement of what diffusers' resnet / sampler /
    attention checkpoints compute.  The vendored resnet has no time-embedding term and norm eps 1e-6: the oracle's `resnet` is called with a zero
    `time_emb_proj` and eps 1e-6, which pins the norm -> SiLU -> conv order, the GroupNorm(32) statistics and the 1x1 shortcut; the `+ time_emb_proj(SiLU(emb))`
    add stays restated (one line).  The vendored Downsample pads (0,1,0,1) and convolves with padding 0 (the VAE form); the UNet's Downsample2D is the
    symmetric padding-1 form: the fixture pins the stride-2 3x3 convolution itself, the padding choice stays restated.
  * Timesteps / TimestepEmbedding lifted from helpers/models/heartmula/codec/transformer.py:15-25, 410-440 (the sinusoid with flip_sin_to_cos and the
    Linear -> SiLU -> Linear embedder) against `timestep_proj` + the oracle's `time_embedding` / `add_embedding` arithmetic.
  * BasicTransformerBlock (norm_type "layer_norm", GEGLU feed-forward, cross-attention over a text context) and Transformer2DModel's GroupNorm -> proj_in ->
    blocks -> proj_out -> residual wiring against tools/ref_shim.py's leaves — an INDEPENDENT restatement of the public diffusers modules (the reference
    carries no copy of these two), marked "shim" in the fixture: it guards the oracle against transcription slips, it is not reference code.
Every case stores inputs, the diffusers-named parameters, the output and d(sum(out * w)) / d(inputs, parameters); tests/test_ref_unet_leaves_cpu.py requires
oracle/unet.py to reproduce them to <= 1e-5 (fp32).  /root/reference is read ONLY here."""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tools import ref_shim  # noqa: E402

OUT = ROOT / "tests" / "golden"


def rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


def run(fn, inputs: dict, params: dict, wseed: int):
    leaves = {k: v.clone().requires_grad_(True) for k, v in {**inputs, **params}.items()}
    out = fn({k: leaves[k] for k in inputs}, {k: leaves[k] for k in params})
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(wseed))
    (out * w).sum().backward()
    return {"inputs": inputs, "params": params, "out": out.detach(), "w": w,
            "grads": {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}}


def load_vendored(module, sd_diffusers, A, strip):
    conv = A.convert_diffusers_state_dict(sd_diffusers)
    missing, unexpected = module.load_state_dict({k[len(strip):]: v for k, v in conv.items()}, strict=True)
    assert not missing and not unexpected
    return module


def functional(module, names):
    """call a vendored nn.Module with an explicit parameter dict (so autograd reaches the fixture's leaves)"""
    from torch.func import functional_call

    def f(x, P):
        return functional_call(module, {n: P[k] for n, k in names.items()}, (x,))
    return f


def main():
    ref_shim.install()
    A = ref_shim.ref_module("simpletuner.helpers.models.ideogram.autoencoder")
    g = torch.Generator().manual_seed(20240)
    cases = {}

    # ---- ResnetBlock: same width, and widening with the 1x1 shortcut -----------------------------------------------------------------------------------
    for tag, ci, co in (("resnet_same", 32, 32), ("resnet_widen", 32, 64)):
        pre = "encoder.mid_block.resnets.0."
        P = {pre + "norm1.weight": 1 + rnd(g, ci, scale=0.2), pre + "norm1.bias": rnd(g, ci, scale=0.2), pre + "conv1.weight": rnd(g, co, ci, 3, 3, scale=0.05),
             pre + "conv1.bias": rnd(g, co, scale=0.1), pre + "norm2.weight": 1 + rnd(g, co, scale=0.2), pre + "norm2.bias": rnd(g, co, scale=0.2),
             pre + "conv2.weight": rnd(g, co, co, 3, 3, scale=0.05), pre + "conv2.bias": rnd(g, co, scale=0.1)}
        if ci != co:
            P[pre + "conv_shortcut.weight"] = rnd(g, co, ci, 1, 1, scale=0.1)
            P[pre + "conv_shortcut.bias"] = rnd(g, co, scale=0.1)
        mod = A.ResnetBlock(ci, co)
        conv_names = {k[len("encoder.mid.block_1."):]: src for src in P for k in [A._rewrite_diffusers_key(src)]}
        f = functional(mod, conv_names)
        cases[tag] = run(lambda i, p: f(i["x"], p), {"x": rnd(g, 2, ci, 8, 8)}, P, 1)
        cases[tag]["kind"] = "reference (ideogram/autoencoder.py:59-87 through convert_diffusers_state_dict :321-392)"

    # ---- Upsample / Downsample ------------------------------------------------------------------------------------------------------------------------
    c = 32
    P = {"decoder.up_blocks.0.upsamplers.0.conv.weight": rnd(g, c, c, 3, 3, scale=0.05), "decoder.up_blocks.0.upsamplers.0.conv.bias": rnd(g, c, scale=0.1)}
    mod = A.Upsample(c)
    f = functional(mod, {"conv.weight": "decoder.up_blocks.0.upsamplers.0.conv.weight", "conv.bias": "decoder.up_blocks.0.upsamplers.0.conv.bias"})
    assert A._rewrite_diffusers_key("decoder.up_blocks.0.upsamplers.0.conv.weight").endswith("upsample.conv.weight")
    cases["upsample"] = run(lambda i, p: f(i["x"], p), {"x": rnd(g, 2, c, 6, 6)}, P, 2)
    cases["upsample"]["kind"] = "reference (ideogram/autoencoder.py:101-110)"
    P = {"encoder.down_blocks.0.downsamplers.0.conv.weight": rnd(g, c, c, 3, 3, scale=0.05), "encoder.down_blocks.0.downsamplers.0.conv.bias": rnd(g, c, scale=0.1)}
    mod = A.Downsample(c)
    f = functional(mod, {"conv.weight": "encoder.down_blocks.0.downsamplers.0.conv.weight", "conv.bias": "encoder.down_blocks.0.downsamplers.0.conv.bias"})
    cases["downsample_vae_padding"] = run(lambda i, p: f(i["x"], p), {"x": rnd(g, 2, c, 8, 8)}, P, 3)
    cases["downsample_vae_padding"]["kind"] = "reference (ideogram/autoencoder.py:89-99): pad (0,1,0,1) + stride-2 conv, padding 0"

    # ---- AttnBlock: GroupNorm(1e-6) -> q / k / v -> single-head attention -> proj_out -> + x ------------------------------------------------------------
    c = 32
    pre = "encoder.mid_block.attentions.0."
    P = {pre + "group_norm.weight": 1 + rnd(g, c, scale=0.2), pre + "group_norm.bias": rnd(g, c, scale=0.2)}
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        P[pre + n + ".weight"] = rnd(g, c, c, scale=0.1)
        P[pre + n + ".bias"] = rnd(g, c, scale=0.1)
    mod = A.AttnBlock(c)

    def attn_ref(i, p):
        from torch.func import functional_call
        conv = A.convert_diffusers_state_dict(p)                 # 2-D linear weights -> 1x1 conv weights (the reference's own reshaping)
        return functional_call(mod, {k[len("encoder.mid.attn_1."):]: v for k, v in conv.items()}, (i["x"],))
    cases["attn_block"] = run(attn_ref, {"x": rnd(g, 2, c, 6, 6)}, P, 4)
    cases["attn_block"]["kind"] = "reference (ideogram/autoencoder.py:29-57 through convert_diffusers_state_dict)"

    # ---- Timesteps + TimestepEmbedding (lifted) --------------------------------------------------------------------------------------------------------
    c0, te = 32, 64
    proj = ref_shim.Timesteps(num_channels=c0, flip_sin_to_cos=True, downscale_freq_shift=0)
    emb = ref_shim.TimestepEmbedding(in_channels=c0, time_embed_dim=te)
    P = {"time_embedding.linear_1.weight": rnd(g, te, c0, scale=0.2), "time_embedding.linear_1.bias": rnd(g, te, scale=0.1),
         "time_embedding.linear_2.weight": rnd(g, te, te, scale=0.2), "time_embedding.linear_2.bias": rnd(g, te, scale=0.1)}
    t = torch.tensor([3.0, 499.0, 871.5])

    def temb_ref(i, p):
        from torch.func import functional_call
        return functional_call(emb, {k[len("time_embedding."):]: v for k, v in p.items()}, (proj(i["t"]),))
    cases["timestep_embedding"] = run(temb_ref, {"t": t}, P, 5)
    cases["timestep_embedding"]["sinusoid"] = proj(t).detach()
    cases["timestep_embedding"]["kind"] = "reference, lifted (heartmula/codec/transformer.py:15-25, 410-440)"

    # ---- BasicTransformerBlock (layer_norm / GEGLU / cross-attention) and the Transformer2DModel wiring: tools/ref_shim.py leaves (independent restatement) ----
    C, heads, ctxd, S_txt = 64, 2, 48, 7
    blk = ref_shim.BasicTransformerBlock(C, heads, C // heads, cross_attention_dim=ctxd, activation_fn="geglu", norm_type="layer_norm")
    pre = "transformer_blocks.0."
    P = {}
    for n, q in blk.named_parameters():
        P[pre + n] = rnd(g, *q.shape, scale=0.1) + (1.0 if n.startswith("norm") and n.endswith("weight") else 0.0)

    def blk_ref(i, p):
        from torch.func import functional_call
        return functional_call(blk, {k[len(pre):]: v for k, v in p.items()}, (i["h"],), {"encoder_hidden_states": i["ctx"]})
    cases["basic_block_shim"] = run(blk_ref, {"h": rnd(g, 2, 36, C), "ctx": rnd(g, 2, S_txt, ctxd)}, P, 6)
    cases["basic_block_shim"].update(kind="shim (independent restatement of diffusers BasicTransformerBlock / GEGLU / Attention)", heads=heads)

    torch.save({"cases": cases, "_cite": "simpletuner/helpers/models/ideogram/autoencoder.py:29-110, :321-392; simpletuner/helpers/models/heartmula/codec/transformer.py:15-25, 410-440"},
               OUT / "ref_unet_leaves.pt")
    for k, v in cases.items():
        print(f"{k:28s} out {tuple(v['out'].shape)}  |out| {float(v['out'].norm()):.4f}  [{v['kind'][:60]}]")


if __name__ == "__main__":
    main()
