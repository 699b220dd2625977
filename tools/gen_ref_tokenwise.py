#!/usr/bin/env python3
"""Generate tests/golden/ref_tokenwise.pt by EXECUTING THE REFERENCE'S OWN MODEL FILES with TOKENWISE timesteps ([B, S_img]: one timestep per image token;
CREPA self-flow, reference tests tests/test_sd3_model.py:179-204, tests/test_flux_model.py:213-272, tests/test_pixart_model.py:91-115).

    python tools/gen_ref_tokenwise.py

Same machinery as tools/gen_ref_models.py (tools/ref_shim.py makes the reference's transformer modules importable unmodified): the reference's model CLASS
gets seeded parameters, runs forward + torch autograd on seeded inputs; stored: inputs, output, d(sum(output * w)) / d(parameters, inputs).
tests/test_ref_models_cpu.py pins the oracle's tokenwise branch to these at <= 1e-5 (fp32).  /root/reference is read ONLY here, never at test time."""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests.ref_fixture_utils import state_checksum  # noqa: E402
from tools import ref_shim  # noqa: E402
from tools.gen_ref_models import run, seed_params, strip  # noqa: E402

OUT = ROOT / "tests" / "golden"


def gen_sd3():
    T = ref_shim.ref_module("simpletuner.helpers.models.sd3.transformer")

    def call(m, a):
        return m(hidden_states=a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], pooled_projections=a["pooled_projections"],
                 timestep=a["timestep"], return_dict=False)[0]

    cfg = dict(sample_size=8, patch_size=2, in_channels=4, num_layers=4, attention_head_dim=16, num_attention_heads=2, joint_attention_dim=24,
               caption_projection_dim=32, pooled_projection_dim=20, out_channels=4, pos_embed_max_size=12)
    model = T.SD3Transformer2DModel(**cfg)
    st = seed_params(model, 411)
    model.eval()
    g = torch.Generator().manual_seed(412)
    B, Hl, Wl = 2, 8, 12
    Si = (Hl // 2) * (Wl // 2)
    inputs = {"hidden_states": torch.randn(B, 4, Hl, Wl, generator=g), "encoder_hidden_states": torch.randn(B, 7, 24, generator=g),
              "pooled_projections": torch.randn(B, 20, generator=g), "timestep": torch.rand(B, Si, generator=g) * 1000.0}
    r = strip(run(model, call, inputs, 413))
    r["inputs"] = inputs
    # the same weights with every token of a sample at that sample's timestep must reproduce the batch-wise forward (the two code paths of the reference agree)
    flat = dict(inputs, timestep=torch.tensor([137.0, 842.0]))
    tok = dict(inputs, timestep=flat["timestep"][:, None].expand(B, Si).contiguous())
    with torch.no_grad():
        a, b = call(model, flat), call(model, tok)
    assert (a - b).norm() / a.norm() < 1e-5, "reference: a constant tokenwise timestep differs from the batch-wise forward"
    return {"config": cfg, "seed": 411, "state_checksum": state_checksum(st), "pos_embed_table": model.pos_embed.pos_embed.detach().clone(), "case": r,
            "_cite": "simpletuner/helpers/models/sd3/transformer.py:61-75 (_sd3_tokenwise_conditioning), :126-142 (AdaLN with a [B, S, D] embedding), :625-626, :680-685, :876"}


def gen_flux():
    T = ref_shim.ref_module("simpletuner.helpers.models.flux.transformer")
    (prepare_latent_image_ids,) = ref_shim.lift(ref_shim.REF / "helpers/models/flux/__init__.py", ["prepare_latent_image_ids"])

    def call(m, a):
        return m(hidden_states=a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], pooled_projections=a["pooled_projections"],
                 timestep=a["timestep"], img_ids=a["img_ids"], txt_ids=a["txt_ids"], guidance=a["guidance"], return_dict=False)[0]

    cfg = dict(patch_size=1, in_channels=16, num_layers=2, num_single_layers=3, attention_head_dim=16, num_attention_heads=2,
               joint_attention_dim=24, pooled_projection_dim=12, guidance_embeds=True, axes_dims_rope=(4, 6, 6))
    model = T.FluxTransformer2DModel(**cfg)
    st = seed_params(model, 421)
    model.eval()
    g = torch.Generator().manual_seed(422)
    B, Hl, Wl, Tt = 2, 8, 12, 5
    S = (Hl // 2) * (Wl // 2)
    inputs = {"hidden_states": torch.randn(B, S, 16, generator=g), "encoder_hidden_states": torch.randn(B, Tt, 24, generator=g),
              "pooled_projections": torch.randn(B, 12, generator=g), "timestep": torch.rand(B, S, generator=g),
              "img_ids": prepare_latent_image_ids(B, Hl, Wl, "cpu", torch.float32), "txt_ids": torch.zeros(Tt, 3), "guidance": torch.tensor([1.0, 3.5])}
    r = strip(run(model, call, inputs, 423))
    r["inputs"] = inputs
    flat = dict(inputs, timestep=torch.tensor([0.137, 0.842]))
    tok = dict(inputs, timestep=flat["timestep"][:, None].expand(B, S).contiguous())
    with torch.no_grad():
        a, b = call(model, flat), call(model, tok)
    assert (a - b).norm() / a.norm() < 1e-5, "reference: a constant tokenwise timestep differs from the batch-wise forward"
    return {"config": cfg, "seed": 421, "state_checksum": state_checksum(st), "latent_hw": (Hl, Wl), "case": r,
            "_cite": "simpletuner/helpers/models/flux/transformer.py:245-294 (_flux_tokenwise_conditioning), :386-412 (AdaLN with a [B, S, D] embedding), :1068-1086 (temb_img / temb_txt / temb_single), :1505"}


def gen_pixart():
    T = ref_shim.ref_module("simpletuner.helpers.models.pixart.transformer")

    def call(m, a):
        return m(a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], timestep=a["timestep"],
                 added_cond_kwargs={"resolution": a["resolution"], "aspect_ratio": a["aspect_ratio"]}, encoder_attention_mask=a["encoder_attention_mask"],
                 return_dict=False)[0]

    cfg = dict(num_attention_heads=2, attention_head_dim=24, in_channels=4, out_channels=8, num_layers=3, cross_attention_dim=48, sample_size=16,
               patch_size=2, caption_channels=20, use_additional_conditions=True)
    model = T.PixArtTransformer2DModel(**cfg)
    st = seed_params(model, 431)
    model.eval()
    g = torch.Generator().manual_seed(432)
    B, Hl, Wl, L = 2, 8, 12, 6
    S = (Hl // 2) * (Wl // 2)
    mask = torch.ones(B, L)
    mask[0, L - 2:] = 0
    inputs = {"hidden_states": torch.randn(B, 4, Hl, Wl, generator=g), "encoder_hidden_states": torch.randn(B, L, 20, generator=g),
              "timestep": torch.rand(B, S, generator=g) * 1000.0, "resolution": torch.tensor([[float(Hl * 8), float(Wl * 8)]] * B),
              "aspect_ratio": torch.tensor([[float(Hl) / float(Wl)]] * B), "encoder_attention_mask": mask}
    r = strip(run(model, call, inputs, 433))
    r["inputs"] = inputs
    flat = dict(inputs, timestep=torch.tensor([137.0, 842.0]))
    tok = dict(inputs, timestep=flat["timestep"][:, None].expand(B, S).contiguous())
    with torch.no_grad():
        a, b = call(model, flat), call(model, tok)
    assert (a - b).norm() / a.norm() < 1e-5, "reference: a constant tokenwise timestep differs from the batch-wise forward"
    return {"config": cfg, "seed": 431, "state_checksum": state_checksum(st), "case": r,
            "_cite": "simpletuner/helpers/models/pixart/transformer.py:60-145 (tokenwise block), :749-753 (head), :790-850 (_embed_timesteps)"}


def gen_pixart_tread():
    """the reference's PixArt trunk with a TREAD route (pixart/transformer.py:487-489, 588-612, 677-741): the router's permutation recorded for replay"""
    from tools.gen_ref_models import RecordingRouter
    T = ref_shim.ref_module("simpletuner.helpers.models.pixart.transformer")
    tread = ref_shim.ref_module("simpletuner.helpers.training.tread")

    def call(m, a):
        return m(a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], timestep=a["timestep"],
                 added_cond_kwargs={"resolution": a["resolution"], "aspect_ratio": a["aspect_ratio"]}, encoder_attention_mask=a["encoder_attention_mask"],
                 return_dict=False)[0]

    cfg = dict(num_attention_heads=2, attention_head_dim=24, in_channels=4, out_channels=8, num_layers=4, cross_attention_dim=48, sample_size=16,
               patch_size=2, caption_channels=20, use_additional_conditions=True)
    model = T.PixArtTransformer2DModel(**cfg)
    st = seed_params(model, 441)
    g = torch.Generator().manual_seed(442)
    B, Hl, Wl, L = 2, 8, 12, 6
    mask = torch.ones(B, L)
    mask[1, L - 1:] = 0
    inputs = {"hidden_states": torch.randn(B, 4, Hl, Wl, generator=g), "encoder_hidden_states": torch.randn(B, L, 20, generator=g),
              "timestep": torch.tensor([137.0, 842.0]), "resolution": torch.tensor([[float(Hl * 8), float(Wl * 8)]] * B),
              "aspect_ratio": torch.tensor([[float(Hl) / float(Wl)]] * B), "encoder_attention_mask": mask}
    routes = [{"selection_ratio": 0.5, "start_layer_idx": 1, "end_layer_idx": -2}]
    rr = RecordingRouter(tread.TREADRouter(seed=9, device="cpu"))
    model.set_router(rr, routes)
    model.train()
    r = strip(run(model, call, inputs, 443))
    r["inputs"], r["routes"], r["mask_infos"] = inputs, routes, rr.infos
    assert len(rr.infos) == 1
    return {"config": cfg, "seed": 441, "state_checksum": state_checksum(st), "case": r,
            "_cite": "simpletuner/helpers/models/pixart/transformer.py:487-489 (set_router), :588-612, :677-741 (the routed span of the block loop)"}


def gen_pixart_lora():
    """peft LoRA on the PixArt trunk's attention projections (pixart/model.py:59), as the reference trains it: the executed model carries the MERGED weights
    W' = W + s B A; the adapter gradients are the ones dL/dW' implies (dA = s B^T dW', dB = s dW' A^T) — the same derivation as the Flux / SD3 tiers of gen_ref_models.py"""
    from tests.ref_fixture_utils import seeded_lora
    from tools.gen_ref_models import lora_grads, merge_lora
    T = ref_shim.ref_module("simpletuner.helpers.models.pixart.transformer")

    def call(m, a):
        return m(a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], timestep=a["timestep"],
                 added_cond_kwargs={"resolution": a["resolution"], "aspect_ratio": a["aspect_ratio"]}, encoder_attention_mask=a["encoder_attention_mask"],
                 return_dict=False)[0]

    cfg = dict(num_attention_heads=2, attention_head_dim=24, in_channels=4, out_channels=8, num_layers=3, cross_attention_dim=48, sample_size=16,
               patch_size=2, caption_channels=20, use_additional_conditions=True)
    model = T.PixArtTransformer2DModel(**cfg)
    st = seed_params(model, 451)
    model.eval()
    shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
    g = torch.Generator().manual_seed(452)
    B, Hl, Wl, L = 2, 8, 12, 6
    mask = torch.ones(B, L)
    mask[0, L - 2:] = 0
    inputs = {"hidden_states": torch.randn(B, 4, Hl, Wl, generator=g), "encoder_hidden_states": torch.randn(B, L, 20, generator=g),
              "timestep": torch.tensor([137.0, 842.0]), "resolution": torch.tensor([[float(Hl * 8), float(Wl * 8)]] * B),
              "aspect_ratio": torch.tensor([[float(Hl) / float(Wl)]] * B), "encoder_attention_mask": mask}
    targets = [f"transformer_blocks.{i}.{a}.{n}" for i in range(3) for a in ("attn1", "attn2") for n in ("to_q", "to_k", "to_v", "to_out.0")]
    rank, alpha = 4, 8.0
    lora = seeded_lora(targets, shapes, rank, 453)
    merge_lora(model, lora, alpha / rank)
    r = run(model, call, inputs, 454)
    return {"config": cfg, "seed": 451, "state_checksum": state_checksum(st), "inputs": inputs, "lora_seed": 453, "lora_rank": rank, "lora_alpha": alpha, "lora_targets": targets,
            "out": r["out"], "w": r["w"], "input_grads": r["input_grads"], "lora_grads": lora_grads(r["_full_grads"], lora, alpha / rank),
            "_cite": "simpletuner/helpers/models/pixart/model.py:59 (DEFAULT_LORA_TARGET); peft LoraLayer: W' = W + (alpha / r) B A"}


if __name__ == "__main__":
    G = {"sd3": gen_sd3(), "flux": gen_flux(), "pixart": gen_pixart(), "pixart_tread": gen_pixart_tread(), "pixart_lora": gen_pixart_lora()}
    torch.save(G, OUT / "ref_tokenwise.pt")
    print({k: (tuple(v["case"]["out"].shape), len(v["case"]["grads"])) for k, v in G.items() if "case" in v}, {"pixart_lora": len(G["pixart_lora"]["lora_grads"])})
