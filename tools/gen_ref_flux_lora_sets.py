#!/usr/bin/env python3
"""Generate tests/golden/ref_flux_lora_sets.pt: the reference's `flux_lora_target` sets beyond the attention projections, on the EXECUTED reference transformer.

    python tools/gen_ref_flux_lora_sets.py

  * the layer lists are READ from the reference's own source (simpletuner/helpers/models/flux/model.py:1235-1380, `get_lora_target_layers`: the `elif
    self.config.flux_lora_target == "<name>": return [...]` chain is walked with `ast`, nothing is typed over) and matched against the reference model's module names
    the way peft's `target_modules` does (a module is wrapped when its name equals an entry or ends with "." + entry; only nn.Linear modules);
  * the reference's FluxTransformer2DModel (tools/ref_shim.py makes it importable unmodified) runs with the MERGED weights W' = W + (alpha / r) B A of seeded
    adapters on exactly those modules; the adapter gradients are the ones dL/dW' implies (dA = s B^T dW', dB = s dW' A^T) — the derivation of tools/gen_ref_models.py.

tests/test_ref_models_cpu.py pins oracle.flux (adapters as separate factors on the same modules, `lora_targets(cfg, which)`) to these at <= 1e-5.
/root/reference is read ONLY here, never at test time."""
from __future__ import annotations

import ast
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests.ref_fixture_utils import seeded_lora, state_checksum  # noqa: E402
from tools import ref_shim  # noqa: E402
from tools.gen_ref_models import lora_grads, merge_lora, run, seed_params  # noqa: E402

OUT = ROOT / "tests" / "golden"


def reference_target_lists() -> dict:
    """{flux_lora_target value: [layer-name suffixes]} from the reference's get_lora_target_layers"""
    src = (ref_shim.REF / "helpers/models/flux/model.py").read_text()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "get_lora_target_layers")
    out = {}
    for node in ast.walk(fn):
        if not (isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and len(node.test.comparators) == 1):
            continue
        left, right = node.test.left, node.test.comparators[0]
        if not (isinstance(left, ast.Attribute) and left.attr == "flux_lora_target" and isinstance(right, ast.Constant) and isinstance(right.value, str)):
            continue
        ret = next((s for s in node.body if isinstance(s, ast.Return) and isinstance(s.value, ast.List)), None)
        if ret is not None:
            out[right.value] = [e.value for e in ret.value.elts if isinstance(e, ast.Constant)]
    return out


def wrapped_modules(model, suffixes) -> list:
    """peft's target_modules rule over the reference model's Linear modules, in module order"""
    return [name for name, m in model.named_modules()
            if isinstance(m, torch.nn.Linear) and any(name == s or name.endswith("." + s) for s in suffixes)]


def gen(which: str, suffixes, layers: int, single: int, seed: int):
    T = ref_shim.ref_module("simpletuner.helpers.models.flux.transformer")
    (prepare_latent_image_ids,) = ref_shim.lift(ref_shim.REF / "helpers/models/flux/__init__.py", ["prepare_latent_image_ids"])

    def call(m, a):
        return m(hidden_states=a["hidden_states"], encoder_hidden_states=a["encoder_hidden_states"], pooled_projections=a["pooled_projections"],
                 timestep=a["timestep"], img_ids=a["img_ids"], txt_ids=a["txt_ids"], guidance=a["guidance"], return_dict=False)[0]

    cfg = dict(patch_size=1, in_channels=16, num_layers=layers, num_single_layers=single, attention_head_dim=16, num_attention_heads=2,
               joint_attention_dim=24, pooled_projection_dim=12, guidance_embeds=True, axes_dims_rope=(4, 6, 6))
    model = T.FluxTransformer2DModel(**cfg)
    st = seed_params(model, seed)
    model.eval()
    shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
    targets = wrapped_modules(model, suffixes)
    assert targets, (which, suffixes)
    rank, alpha = 4, 8.0
    lora = seeded_lora(targets, shapes, rank, seed + 1)
    merge_lora(model, lora, alpha / rank)
    g = torch.Generator().manual_seed(seed + 2)
    B, Hl, Wl, Tt = 2, 8, 12, 5
    S = (Hl // 2) * (Wl // 2)
    inputs = {"hidden_states": torch.randn(B, S, 16, generator=g), "encoder_hidden_states": torch.randn(B, Tt, 24, generator=g),
              "pooled_projections": torch.randn(B, 12, generator=g), "timestep": torch.tensor([0.137, 0.842]),
              "img_ids": prepare_latent_image_ids(B, Hl, Wl, "cpu", torch.float32), "txt_ids": torch.zeros(Tt, 3), "guidance": torch.tensor([1.0, 3.5])}
    r = run(model, call, inputs, seed + 3)
    return {"config": cfg, "seed": seed, "state_checksum": state_checksum(st), "inputs": inputs, "lora_seed": seed + 1, "lora_rank": rank, "lora_alpha": alpha,
            "suffixes": list(suffixes), "lora_targets": targets, "out": r["out"], "w": r["w"], "input_grads": r["input_grads"],
            "lora_grads": lora_grads(r["_full_grads"], lora, alpha / rank)}


if __name__ == "__main__":
    lists = reference_target_lists()
    print({k: len(v) for k, v in lists.items()})
    G = {"_cite": "simpletuner/helpers/models/flux/model.py:1235-1380 (get_lora_target_layers); peft LoraLayer: W' = W + (alpha / r) B A on the modules target_modules names",
         "all+ffs": gen("all+ffs", lists["all+ffs"], 2, 3, 511), "context+ffs": gen("context+ffs", lists["context+ffs"], 2, 2, 521),
         "context": gen("context", lists["context"], 2, 2, 531), "all": gen("all", lists["all"], 2, 2, 541),
         "nano": gen("nano", lists["nano"], 1, 9, 551), "tiny": gen("tiny", lists["tiny"], 1, 22, 561),
         "all+ffs+embedder": gen("all+ffs+embedder", lists["all+ffs+embedder"], 1, 2, 571),
         "ai-toolkit": gen("ai-toolkit", lists["ai-toolkit"], 2, 2, 581)}
    torch.save(G, OUT / "ref_flux_lora_sets.pt")
    print({k: (len(v["lora_targets"]), tuple(v["out"].shape)) for k, v in G.items() if isinstance(v, dict)})
