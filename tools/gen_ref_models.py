#!/usr/bin/env python3
"""Generate tests/golden/ref_{flux,sd3,pixart}_model.pt by EXECUTING THE REFERENCE'S OWN MODEL FILES in this container.

    python tools/gen_ref_models.py [flux] [sd3] [pixart] [vae]

`tools/ref_shim.py` makes `simpletuner.helpers.models.{flux,sd3,pixart}.transformer` / `pixart.controlnet` importable unmodified (a fake
`diffusers` holding only leaf modules; no reference `__init__` runs).  This script builds the reference's model CLASSES, gives every parameter
a seeded value, runs the reference `forward` + torch autograd on seeded inputs and commits two tiers per family:

  "tiny"  (D = 32..48, several blocks; weights rebuilt from a seed by both sides, checksum stored): inputs, output and d(sum(output * w))/d(parameters, inputs) for: the plain forward, the text-key mask (Flux), SD3.5 (q/k RMSNorm + dual
          attention), non-square latents, TREAD routing (router permutations recorded for replay), plus — per activation-checkpoint mode —
          WHICH blocks the reference wrapped (recorded by intercepting its checkpoint function).  tests/test_ref_models_cpu.py pins
          oracle/{flux,sd3,pixart}.py to these at <= 1e-5 (fp32).
  "hip"   (the head widths the HIP kernels are built for: 2 x 128 Flux, 2 x 64 SD3, 8 x 72 PixArt): NO weights stored — both sides rebuild
          them with tests/ref_fixture_utils.seeded_state (bf16-representable values; checksum stored).  Stored: inputs, output, input gradients
          and, for the adapters, the LoRA gradients the reference's dL/dW implies (peft: W' = W + s B A  =>  dA = s B^T dW', dB = s dW' A^T).
          tests/test_ref_models_gpu.py runs the HIP models against these — the product is compared with executed reference code, not only
          with the repo's own restatement; the CPU suite checks the oracle against the same tier.

/root/reference is read ONLY here, never at test time.
"""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests.ref_fixture_utils import seeded_lora, seeded_state, state_checksum  # noqa: E402
from tools import ref_shim  # noqa: E402

OUT = ROOT / "tests" / "golden"


def seed_params(model: torch.nn.Module, seed: int, bf16: bool = False):
    st = seeded_state({n: tuple(p.shape) for n, p in model.named_parameters()}, seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(st[n].to(torch.bfloat16).float() if bf16 else st[n])
    return {n: p.detach().clone() for n, p in model.named_parameters()}


def bf(t):
    return t.to(torch.bfloat16).float()


def run(model, call, inputs: dict, wseed: int, keep=None):
    """forward + backward of sum(out * w); returns dict(out, w, grads{param}, input_grads{name}); `keep`: name prefixes of the param grads kept"""
    for p in model.parameters():
        p.grad = None
    leaves = {k: v.clone().requires_grad_(True) for k, v in inputs.items() if torch.is_tensor(v) and v.is_floating_point()}
    args = dict(inputs)
    args.update(leaves)
    out = call(model, args)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(wseed))
    (out * w).sum().backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    full = grads
    if keep is not None:
        grads = {n: g for n, g in grads.items() if any(n.startswith(k) for k in keep)}
    return {"out": out.detach().clone(), "w": w, "grads": grads, "input_grads": {k: v.grad.detach().clone() for k, v in leaves.items() if v.grad is not None},
            "_full_grads": full}


def strip(r):
    r.pop("_full_grads", None)
    return r


class CheckpointSpy:
    """intercepts a reference module's `simpletuner_checkpoint` and records, per call, which (named) modules ran inside it"""

    def __init__(self, mod, named):
        self.mod, self.named, self.calls, self.inside, self.hooks = mod, named, [], None, []
        self.real = mod.simpletuner_checkpoint

    def __enter__(self):
        spy = self

        def ckpt(fn, *a, **k):
            spy.inside = []
            spy.calls.append(spy.inside)
            try:
                return spy.real(fn, *a, **k)
            finally:
                spy.inside = None

        self.mod.simpletuner_checkpoint = ckpt
        for name, m in self.named.items():
            def pre(_m, _args, _n=name):
                if spy.inside is not None and _n not in spy.inside:
                    spy.inside.append(_n)
            self.hooks.append(m.register_forward_pre_hook(pre))
        return self

    def __exit__(self, *a):
        self.mod.simpletuner_checkpoint = self.real
        for h in self.hooks:
            h.remove()


class RecordingRouter:
    """the reference TREADRouter with every MaskInfo it hands out recorded (so the product path can replay the permutations)"""

    def __init__(self, router):
        self.router, self.infos = router, []

    def get_mask(self, *a, **k):
        info = self.router.get_mask(*a, **k)
        self.infos.append({"mask": info.mask.clone(), "ids_keep": info.ids_keep.clone(), "ids_mask": info.ids_mask.clone(),
                           "ids_shuffle": info.ids_shuffle.clone(), "ids_restore": info.ids_restore.clone()})
        return info

    def start_route(self, *a, **k):
        return self.router.start_route(*a, **k)

    def end_route(self, *a, **k):
        return self.router.end_route(*a, **k)


def merge_lora(model, lora, scale):
    with torch.no_grad():
        for name, (A, B) in lora.items():
            w = model.get_parameter(name + ".weight")
            w.add_(scale * (B @ A))


def lora_grads(full_grads, lora, scale):
    out = {}
    for name, (A, B) in lora.items():
        dW = full_grads[name + ".weight"]
        out[name] = (scale * (B.t() @ dW), scale * (dW @ A.t()))
    return out


def checkpoint_plans(T, model, named, call, inputs, ref_out, modes):
    plans = {}
    model.train()
    model.gradient_checkpointing = True
    for tag, interval, stride in modes:
        model.set_gradient_checkpointing_interval(interval)
        model.set_gradient_checkpointing_segment_stride(stride)
        with CheckpointSpy(T, named) as spy:
            r = run(model, call, inputs, 303)
        plans[tag] = {"interval": interval, "stride": stride, "wrapped": [list(c) for c in spy.calls]}
        assert torch.equal(r["out"], ref_out), tag
    model.gradient_checkpointing = False
    model.set_gradient_checkpointing_interval(None)
    model.set_gradient_checkpointing_segment_stride(None)
    model.eval()
    return plans


MODES = (("layer", None, None), ("interval2", 2, None), ("interval3", 3, None), ("seg2_stride3", 2, 3), ("seg2_stride4", 2, 4))


# ------------------------------------------------------------------------------------------------------------------------
def gen_flux():
    T = ref_shim.ref_module("simpletuner.helpers.models.flux.transformer")
    tread = ref_shim.ref_module("simpletuner.helpers.training.tread")
    pack_latents, prepare_latent_image_ids = ref_shim.lift(ref_shim.REF / "helpers/models/flux/__init__.py", ["pack_latents", "prepare_latent_image_ids"])

    def call(m, a, **extra):
        return m(hidden_states=a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], pooled_projections=a["pooled_projections"],
                 timestep=a["timestep"], img_ids=a["img_ids"], txt_ids=a["txt_ids"], guidance=a["guidance"], return_dict=False, **extra)[0]

    # ---- tiny tier ----
    cfg = dict(patch_size=1, in_channels=16, num_layers=3, num_single_layers=5, attention_head_dim=16, num_attention_heads=2,
               joint_attention_dim=24, pooled_projection_dim=12, guidance_embeds=True, axes_dims_rope=(4, 6, 6))
    model = T.FluxTransformer2DModel(**cfg)
    state = seed_params(model, 101)
    model.eval()
    g = torch.Generator().manual_seed(202)
    B, Hl, Wl, Tt = 2, 8, 12, 5                      # latent 8 x 12 -> 4 x 6 = 24 packed tokens of 16 channels
    S = (Hl // 2) * (Wl // 2)
    inputs = {
        "hidden_states": torch.randn(B, S, 16, generator=g),
        "encoder_hidden_states": torch.randn(B, Tt, 24, generator=g),
        "pooled_projections": torch.randn(B, 12, generator=g),
        "timestep": torch.tensor([0.137, 0.842]),
        "img_ids": prepare_latent_image_ids(B, Hl, Wl, "cpu", torch.float32),
        "txt_ids": torch.zeros(Tt, 3),
        "guidance": torch.tensor([1.0, 3.5]),
    }
    tiny = {"config": cfg, "seed": 101, "state_checksum": state_checksum(state), "inputs": inputs, "latent_hw": (Hl, Wl), "cases": {}}
    tiny["cases"]["plain"] = strip(run(model, call, inputs, 303))
    sub = ("transformer_blocks.1.", "single_transformer_blocks.1.", "x_embedder", "norm_out")
    mask = torch.tensor([[1.0, 1.0, 1.0, 0.0, 0.0], [1.0, 1.0, 1.0, 1.0, 1.0]])
    tiny["cases"]["masked"] = strip(run(model, lambda m, a: call(m, a, attention_mask=mask), inputs, 304, keep=sub))
    tiny["cases"]["masked"]["attention_mask"] = mask
    named = {f"d{i}": b for i, b in enumerate(model.transformer_blocks)}
    named.update({f"s{i}": b for i, b in enumerate(model.single_transformer_blocks)})
    tiny["checkpoint_plans"] = checkpoint_plans(T, model, named, call, inputs, tiny["cases"]["plain"]["out"], MODES)
    for tag, routes in (("tread_double", [{"selection_ratio": 0.5, "start_layer_idx": 1, "end_layer_idx": 2}]),
                        ("tread_single", [{"selection_ratio": 0.25, "start_layer_idx": 4, "end_layer_idx": -2}])):
        rr = RecordingRouter(tread.TREADRouter(seed=7, device="cpu"))
        model.set_router(rr, routes)
        model.train()
        r = strip(run(model, call, inputs, 305, keep=sub))
        r["routes"], r["mask_infos"] = routes, rr.infos
        tiny["cases"][tag] = r
    model.set_router(None, None)

    # ---- hip tier: 2 heads x 128, 2 double + 2 single blocks, LoRA r4 alpha 8 on the reference's default targets ----
    hcfg = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
                joint_attention_dim=64, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
    model = T.FluxTransformer2DModel(**hcfg)
    st = seed_params(model, 141, bf16=True)
    model.eval()
    shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
    targets = [f"transformer_blocks.{i}.attn.{n}" for i in range(2) for n in ("to_q", "to_k", "to_v", "to_out.0")] + \
              [f"single_transformer_blocks.{i}.attn.{n}" for i in range(2) for n in ("to_q", "to_k", "to_v")]
    rank, alpha = 4, 8.0
    lora = seeded_lora(targets, shapes, rank, 151)
    merge_lora(model, lora, alpha / rank)
    g = torch.Generator().manual_seed(242)
    B, Hl, Wl, Tt = 2, 16, 24, 32
    latents = bf(torch.randn(B, 16, Hl, Wl, generator=g))
    hin = {
        "hidden_states": pack_latents(latents, B, 16, Hl, Wl),
        "encoder_hidden_states": bf(torch.randn(B, Tt, 64, generator=g)),
        "pooled_projections": bf(torch.randn(B, 64, generator=g)),
        "timestep": torch.tensor([0.25, 0.8125]),
        "img_ids": prepare_latent_image_ids(B, Hl, Wl, "cpu", torch.float32),
        "txt_ids": torch.zeros(Tt, 3),
        "guidance": torch.tensor([1.0, 1.0]),
    }
    r = run(model, call, hin, 343)
    lg = lora_grads(r["_full_grads"], lora, alpha / rank)
    hip = {"config": hcfg, "seed": 141, "lora_seed": 151, "lora_rank": rank, "lora_alpha": alpha, "lora_targets": targets,
           "state_checksum": state_checksum(st), "latents": latents, "latent_hw": (Hl, Wl), "inputs": hin,
           "out": r["out"], "w": r["w"], "input_grads": r["input_grads"], "lora_grads": lg}
    G = {"tiny": tiny, "hip": hip,
         "_cite": ("simpletuner/helpers/models/flux/transformer.py:73-224 (RoPE, FluxAttnProcessor2_0), :386-412 (AdaLN helpers), :415-510 (single block), "
                   ":513-687 (double block), :690-1513 (model forward incl. checkpoint plans and TREAD routing); flux/__init__.py:25-63; "
                   "training/tread.py:58-159; training/gradient_checkpointing_interval.py:48-120")}
    torch.save(G, OUT / "ref_flux_model.pt")
    print("flux tiny:", list(tiny["cases"]), "plans:", {k: v["wrapped"] for k, v in tiny["checkpoint_plans"].items()})
    print("flux hip out", tuple(hip["out"].shape), "checksum", hip["state_checksum"])


def gen_sd3():
    T = ref_shim.ref_module("simpletuner.helpers.models.sd3.transformer")
    tread = ref_shim.ref_module("simpletuner.helpers.training.tread")

    def call(m, a):
        return m(hidden_states=a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], pooled_projections=a["pooled_projections"],
                 timestep=a["timestep"], return_dict=False)[0]

    G = {"tiny": {}, "hip": {}}
    for vtag, extra in (("sd3", {}), ("sd35", {"qk_norm": "rms_norm", "dual_attention_layers": (0, 1)})):
        cfg = dict(sample_size=8, patch_size=2, in_channels=4, num_layers=5, attention_head_dim=16, num_attention_heads=2, joint_attention_dim=24,
                   caption_projection_dim=32, pooled_projection_dim=20, out_channels=4, pos_embed_max_size=12, **extra)
        model = T.SD3Transformer2DModel(**cfg)
        st = seed_params(model, 111)
        model.eval()
        g = torch.Generator().manual_seed(212)
        B = 2
        V = {"config": cfg, "seed": 111, "state_checksum": state_checksum(st), "pos_embed_table": model.pos_embed.pos_embed.detach().clone(), "cases": {}}
        sub = ("transformer_blocks.0.", "transformer_blocks.4.", "pos_embed", "norm_out")
        for ctag, (Hl, Wl) in ((("wide", (8, 12)), ("square", (8, 8)), ("tall", (12, 8))) if vtag == "sd3" else (("wide", (8, 12)),)):
            inputs = {"hidden_states": torch.randn(B, 4, Hl, Wl, generator=g), "encoder_hidden_states": torch.randn(B, 7, 24, generator=g),
                      "pooled_projections": torch.randn(B, 20, generator=g), "timestep": torch.tensor([137.0, 842.0])}
            r = strip(run(model, call, inputs, 313, keep=None if (ctag == "wide" and vtag == "sd3") else sub))
            r["inputs"] = inputs
            V["cases"][ctag] = r
        if vtag == "sd3":
            inputs = V["cases"]["wide"]["inputs"]
            # the per-block path calls _sd3_apply_joint_transformer_block(block, ...) (never block.forward): spy on each block's norm1 instead
            named = {f"b{i}": b.norm1 for i, b in enumerate(model.transformer_blocks)}
            V["checkpoint_plans"] = checkpoint_plans(T, model, named, call, inputs, V["cases"]["wide"]["out"], MODES)
            routes = [{"selection_ratio": 0.5, "start_layer_idx": 1, "end_layer_idx": 2}]
            rr = RecordingRouter(tread.TREADRouter(seed=9, device="cpu"))
            model.set_router(rr, routes)
            model.train()
            r = strip(run(model, call, inputs, 315, keep=sub))
            r["routes"], r["mask_infos"], r["inputs"] = routes, rr.infos, inputs
            V["cases"]["tread"] = r
            model.set_router(None, None)
            model.eval()
            print("sd3 plans:", {k: v["wrapped"] for k, v in V["checkpoint_plans"].items()})
        G["tiny"][vtag] = V

    # ---- hip tier: 2 heads x 64, 3 blocks (last context_pre_only); LoRA r4 gradients AND full-fine-tune gradients of a few tensors ----
    for vtag, extra in (("sd3", {}), ("sd35", {"qk_norm": "rms_norm", "dual_attention_layers": (0,)})):
        hcfg = dict(sample_size=32, patch_size=2, in_channels=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=128,
                    caption_projection_dim=128, pooled_projection_dim=64, out_channels=16, pos_embed_max_size=24, **extra)
        model = T.SD3Transformer2DModel(**hcfg)
        st = seed_params(model, 161, bf16=True)
        model.eval()
        shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
        g = torch.Generator().manual_seed(262)
        B, Hl, Wl, Tt = 2, 16, 24, 24
        hin = {"hidden_states": bf(torch.randn(B, 16, Hl, Wl, generator=g)), "encoder_hidden_states": bf(torch.randn(B, Tt, 128, generator=g)),
               "pooled_projections": bf(torch.randn(B, 64, generator=g)), "timestep": torch.tensor([250.0, 812.5])}
        full = run(model, call, hin, 363)
        keep = ["transformer_blocks.0.attn.to_q.weight", "transformer_blocks.0.attn.add_k_proj.weight", "transformer_blocks.1.ff.net.0.proj.bias",
                "transformer_blocks.1.ff.net.2.bias", "transformer_blocks.1.norm1.linear.bias", "transformer_blocks.2.norm1_context.linear.bias",
                "transformer_blocks.2.attn.to_out.0.weight", "pos_embed.proj.weight", "norm_out.linear.bias", "proj_out.weight", "context_embedder.weight",
                "time_text_embed.timestep_embedder.linear_2.weight"]
        if extra:
            keep += ["transformer_blocks.0.attn.norm_q.weight", "transformer_blocks.0.attn.norm_added_k.weight", "transformer_blocks.0.attn2.to_v.weight",
                     "transformer_blocks.0.attn2.norm_k.weight", "transformer_blocks.0.norm1.linear.bias"]
        H = {"config": hcfg, "seed": 161, "state_checksum": state_checksum(st), "inputs": hin, "out": full["out"], "w": full["w"],
             "input_grads": full["input_grads"], "full_ft_grads": {k: full["_full_grads"][k] for k in keep}}
        if not extra:
            targets = [f"transformer_blocks.{i}.attn.{n}" for i in range(3) for n in ("to_q", "to_k", "to_v", "to_out.0")]
            rank, alpha = 4, 8.0
            lora = seeded_lora(targets, shapes, rank, 171)
            merge_lora(model, lora, alpha / rank)
            r = run(model, call, hin, 363)
            H["lora"] = {"lora_seed": 171, "lora_rank": rank, "lora_alpha": alpha, "lora_targets": targets, "out": r["out"],
                         "input_grads": r["input_grads"], "lora_grads": lora_grads(r["_full_grads"], lora, alpha / rank)}
        G["hip"][vtag] = H
    G["_cite"] = ("simpletuner/helpers/models/sd3/transformer.py:126-142 (AdaLN helpers), :145-241 (_sd3_apply_joint_transformer_block incl. dual attention), "
                  ":560-911 (model forward, checkpoint plans, TREAD, unpatchify)")
    torch.save(G, OUT / "ref_sd3_model.pt")
    print("sd3 tiny:", {v: list(G["tiny"][v]["cases"]) for v in G["tiny"]}, "hip:", list(G["hip"]))


def gen_pixart():
    T = ref_shim.ref_module("simpletuner.helpers.models.pixart.transformer")
    C = ref_shim.ref_module("simpletuner.helpers.models.pixart.controlnet")

    def call_trunk(m, a):
        return m(a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], timestep=a["timestep"],
                 added_cond_kwargs={"resolution": a["resolution"], "aspect_ratio": a["aspect_ratio"]},
                 encoder_attention_mask=a["encoder_attention_mask"], return_dict=False)[0]

    def call_wrap(m, a):
        return m(a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], timestep=a["timestep"], controlnet_cond=a["controlnet_cond"],
                 added_cond_kwargs={"resolution": a["resolution"], "aspect_ratio": a["aspect_ratio"]},
                 encoder_attention_mask=a["encoder_attention_mask"], return_dict=False)[0]

    def make(cfg, n_ctrl, seed, bf16, B, Hl, Wl, L, Cc, gseed):
        trunk = T.PixArtTransformer2DModel(**cfg)
        st = seed_params(trunk, seed, bf16=bf16)
        trunk.eval()
        D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
        adapter = C.PixArtSigmaControlNetAdapterModel(num_layers=n_ctrl, num_attention_heads=cfg["num_attention_heads"],
                                                      attention_head_dim=cfg["attention_head_dim"], cross_attention_dim=D)
        ast_ = seed_params(adapter, seed + 10, bf16=bf16)
        g = torch.Generator().manual_seed(gseed)
        r = (lambda t: bf(t)) if bf16 else (lambda t: t)
        mask = torch.ones(B, L)
        mask[0, L - 2:] = 0
        inputs = {
            "hidden_states": r(torch.randn(B, 4, Hl, Wl, generator=g)),
            "encoder_hidden_states": r(torch.randn(B, L, Cc, generator=g)),
            "timestep": torch.tensor([137.0, 842.0]),
            "resolution": torch.tensor([[float(Hl * 8), float(Wl * 8)]] * B),
            "aspect_ratio": torch.tensor([[float(Hl) / float(Wl)]] * B),
            "encoder_attention_mask": mask,
            "controlnet_cond": r(torch.randn(B, 4, Hl, Wl, generator=g)),
        }
        return trunk, adapter, st, ast_, inputs

    # ---- tiny ----
    cfg = dict(num_attention_heads=2, attention_head_dim=24, in_channels=4, out_channels=8, num_layers=4, cross_attention_dim=48, sample_size=16,
               patch_size=2, caption_channels=20, use_additional_conditions=True)
    trunk, adapter, st, ast_, inputs = make(cfg, 2, 121, False, 2, 8, 12, 6, 20, 222)
    tiny = {"config": cfg, "n_ctrl": 2, "inputs": inputs, "cases": {}, "seed": 121, "adapter_seed": 131, "state_checksum": state_checksum(st),
            "adapter_checksum": state_checksum(ast_)}
    tiny["cases"]["trunk"] = strip(run(trunk, call_trunk, {k: v for k, v in inputs.items() if k != "controlnet_cond"}, 323,
                                       keep=("transformer_blocks.0.", "transformer_blocks.3.", "adaln_single", "caption_projection", "pos_embed", "scale_shift_table", "proj_out")))
    with torch.no_grad():
        # the in-tree tokenwise block (pixart/transformer.py:95-145) on a [B, S, 6D] broadcast of the batch-wise modulation must equal the batch-wise block
        blk, D, B, L = trunk.transformer_blocks[1], 48, 2, 6
        g = torch.Generator().manual_seed(223)
        h, ctx, t6 = torch.randn(B, 24, D, generator=g), torch.randn(B, L, D, generator=g), torch.randn(B, 6 * D, generator=g)
        bias = ((1 - inputs["encoder_attention_mask"]) * -10000.0).unsqueeze(1)
        a = blk(h, encoder_hidden_states=ctx, encoder_attention_mask=bias, timestep=t6)
        b = blk(h, encoder_hidden_states=ctx, encoder_attention_mask=bias, timestep=t6[:, None, :].expand(B, 24, 6 * D).contiguous())
        assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
        tiny["block_case"] = {"h": h, "ctx": ctx, "t6": t6, "bias": bias, "out_tokenwise_reference_code": b, "block_index": 1}
    wrap = C.PixArtSigmaControlNetTransformerModel(trunk, adapter, training=True)
    r = run(wrap, call_wrap, inputs, 325)
    r["grads"] = {k[len("controlnet."):]: v for k, v in r["_full_grads"].items() if k.startswith("controlnet.")}
    tiny["cases"]["controlnet"] = strip(r)

    # ---- hip: 8 heads x 72 (the product needs the inner width to be a multiple of 64), 3 trunk blocks, 2 adapter blocks ----
    hcfg = dict(num_attention_heads=8, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=3, cross_attention_dim=576, sample_size=128,
                patch_size=2, caption_channels=64, use_additional_conditions=True)
    trunk, adapter, st, ast_, hin = make(hcfg, 2, 181, True, 2, 16, 24, 24, 64, 282)
    rt = run(trunk, call_trunk, {k: v for k, v in hin.items() if k != "controlnet_cond"}, 383)
    wrap = C.PixArtSigmaControlNetTransformerModel(trunk, adapter, training=True)
    r = run(wrap, call_wrap, hin, 385)
    keep = ["controlnet_blocks.1.after_proj.weight", "controlnet_blocks.1.after_proj.bias",
            "controlnet_blocks.0.transformer_block.scale_shift_table", "controlnet_blocks.0.transformer_block.attn1.to_q.bias",
            "controlnet_blocks.1.transformer_block.attn2.to_v.bias", "controlnet_blocks.1.transformer_block.ff.net.0.proj.bias",
            "controlnet_blocks.1.transformer_block.ff.net.2.bias", "controlnet_blocks.0.before_proj.bias",
            "controlnet_blocks.1.transformer_block.attn1.to_out.0.bias"]
    hip = {"config": hcfg, "n_ctrl": 2, "seed": 181, "adapter_seed": 191, "state_checksum": state_checksum(st), "adapter_checksum": state_checksum(ast_),
           "inputs": hin, "trunk_out": rt["out"], "trunk_w": rt["w"], "trunk_input_grads": rt["input_grads"],
           "out": r["out"], "w": r["w"], "input_grads": r["input_grads"], "adapter_grads": {k: r["_full_grads"]["controlnet." + k] for k in keep}}
    G = {"tiny": tiny, "hip": hip,
         "_cite": ("simpletuner/helpers/models/pixart/transformer.py:95-145 (tokenwise block), :499-788 (model forward); pixart/controlnet.py:13-98 "
                   "(adapter block), :166-326 (wrapper forward)")}
    torch.save(G, OUT / "ref_pixart_model.pt")
    print("pixart tiny:", {k: tuple(v["out"].shape) for k, v in tiny["cases"].items()}, "hip out", tuple(hip["out"].shape))


def vae_shapes(ch, latent, quant_conv, layers=2):
    """diffusers AutoencoderKL state-dict names -> shapes for block_out_channels `ch` (what a released VAE checkpoint holds; the reference's own
    `convert_diffusers_state_dict` below accepts exactly these names and its `load_state_dict` these shapes)"""
    sh = {}

    def conv(n, ci, co, k=3):
        sh[n + ".weight"], sh[n + ".bias"] = (co, ci, k, k), (co,)

    def norm(n, c):
        sh[n + ".weight"], sh[n + ".bias"] = (c,), (c,)

    def res(p_, ci, co):
        norm(p_ + "norm1", ci); conv(p_ + "conv1", ci, co); norm(p_ + "norm2", co); conv(p_ + "conv2", co, co)
        if ci != co:
            conv(p_ + "conv_shortcut", ci, co, 1)

    def mid(p_, c):
        res(p_ + "resnets.0.", c, c)
        norm(p_ + "attentions.0.group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            sh[p_ + f"attentions.0.{n}.weight"], sh[p_ + f"attentions.0.{n}.bias"] = (c, c), (c,)
        res(p_ + "resnets.1.", c, c)

    conv("encoder.conv_in", 3, ch[0])
    cin = ch[0]
    for i, co in enumerate(ch):
        for j in range(layers):
            res(f"encoder.down_blocks.{i}.resnets.{j}.", cin, co)
            cin = co
        if i < len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cin, cin)
    mid("encoder.mid_block.", cin)
    norm("encoder.conv_norm_out", cin); conv("encoder.conv_out", cin, 2 * latent)
    if quant_conv:
        conv("quant_conv", 2 * latent, 2 * latent, 1); conv("post_quant_conv", latent, latent, 1)
    rev = tuple(reversed(ch))
    conv("decoder.conv_in", latent, rev[0])
    mid("decoder.mid_block.", rev[0])
    cin = rev[0]
    for i, co in enumerate(rev):
        for j in range(layers + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}.", cin, co)
            cin = co
        if i < len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cin, cin)
    norm("decoder.conv_norm_out", cin); conv("decoder.conv_out", cin, 3)
    return sh


def gen_vae():
    """The KL autoencoder the reference vendors WITH a diffusers-key converter (models/ideogram/autoencoder.py: Encoder / Decoder / AttnBlock / ResnetBlock
    in the original latent-diffusion layout + `convert_diffusers_state_dict`): the reference's own statement of what an `AutoencoderKL` checkpoint computes.
    Seeded diffusers-named weights -> its converter -> its `load_state_dict` -> its `encoder(x)` (moments) and `decoder(z)` (pixels)."""
    A = ref_shim.ref_module("simpletuner.helpers.models.ideogram.autoencoder")
    cases = {}
    for name, latent, quant, hw, seed in (("sdxl_layout", 4, True, (32, 48), 301), ("flux_layout", 16, False, (48, 32), 311)):
        ch = (32, 64, 64, 64)
        shapes = vae_shapes(ch, latent, quant)
        st = seeded_state(shapes, seed)
        st = {k: (1.0 + 0.1 * v if (("norm" in k) and k.endswith(".weight")) else v) for k, v in st.items()}       # norm scales around 1
        sd = dict(st)
        if not quant:      # a checkpoint without quant convs (FLUX.1 VAE): the vendored Encoder / Decoder always hold them -> identity there
            sd["quant_conv.weight"] = torch.eye(2 * latent).view(2 * latent, 2 * latent, 1, 1); sd["quant_conv.bias"] = torch.zeros(2 * latent)
            sd["post_quant_conv.weight"] = torch.eye(latent).view(latent, latent, 1, 1); sd["post_quant_conv.bias"] = torch.zeros(latent)
        model = A.AutoEncoder(A.AutoEncoderParams(resolution=hw[0], in_channels=3, ch=32, out_ch=3, ch_mult=[1, 2, 2, 2], num_res_blocks=2, z_channels=latent))
        missing, unexpected = model.load_state_dict(A.convert_diffusers_state_dict(sd), strict=False)
        assert not unexpected and all(k.startswith("bn.") for k in missing), (missing, unexpected)
        model.eval()
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(2, 3, *hw, generator=g).clamp(-1, 1).requires_grad_(True)
        z = torch.randn(2, latent, hw[0] // 8, hw[1] // 8, generator=g).requires_grad_(True)
        w_m = torch.randn(2, 2 * latent, hw[0] // 8, hw[1] // 8, generator=g)
        w_p = torch.randn(2, 3, *hw, generator=g)
        moments = model.encoder(x)
        (moments * w_m).sum().backward()
        pixels = model.decoder(z)
        (pixels * w_p).sum().backward()
        cases[name] = {"block_out_channels": ch, "latent_channels": latent, "use_quant_conv": quant, "seed": seed, "state_checksum": state_checksum(st),
                       "x": x.detach(), "z": z.detach(), "w_m": w_m, "w_p": w_p, "moments": moments.detach(), "pixels": pixels.detach(),
                       "dx": x.grad.detach(), "dz": z.grad.detach()}
    torch.save({"cases": cases, "_cite": "simpletuner/helpers/models/ideogram/autoencoder.py:29-57 (AttnBlock), :59-87 (ResnetBlock), :89-110 (Down/Upsample), "
                                         ":113-186 (Encoder), :189-273 (Decoder), :321-392 (convert_diffusers_state_dict)"}, OUT / "ref_vae_model.pt")
    print("vae:", {k: (tuple(v["moments"].shape), tuple(v["pixels"].shape)) for k, v in cases.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["flux", "sd3", "pixart", "vae"]
    for w_ in which:
        {"flux": gen_flux, "sd3": gen_sd3, "pixart": gen_pixart, "vae": gen_vae}[w_]()
