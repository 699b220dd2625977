#!/usr/bin/env python3
"""Generate tests/golden/ref_unet_walk.pt: the DOWN HALF of the UNet2DConditionModel walk, from reference code EXECUTED in this container.

    python tools/gen_ref_unet_walk.py

The reference imports diffusers' UNet2DConditionModel and never vendors it — but it does vendor diffusers' ControlNetModel
(simpletuner/helpers/models/kolors/controlnet.py:132-931), whose constructor and forward ARE the UNet's up to the mid block: the timestep embedder
(`time_embed_dim = 4 * block_out_channels[0]`), the SDXL "text_time" addition embedding (`time_ids.flatten() -> add_time_proj -> reshape(B, -1) ->
cat([text_embeds, time_embeds]) -> add_embedding`, `emb = emb + aug_emb`), `num_attention_heads = num_attention_heads or attention_head_dim` (the config key
named attention_head_dim is the HEAD COUNT), conv_in, the down-block loop that collects `(sample,) + res_samples`, the mid block.  This script imports THAT FILE
unmodified (tools/ref_shim.py's path-only packages + a fake `diffusers`) and runs it on a small SDXL-shaped configuration.

What is executed and what is a stand-in:
  * executed: ControlNetModel.__init__ / .forward (the wiring and the walk), Timesteps / TimestepEmbedding (lifted by ref_shim from
    helpers/models/heartmula/codec/transformer.py:410-440), torch's Conv2d for conv_in and the 1x1 output convolutions (set to the identity here, so the
    outputs ARE the skip tensors; the conditioning embedding's last convolution keeps its zero initialisation, so it adds nothing);
  * stand-ins: `get_down_block` / CrossAttnDownBlock2D / DownBlock2D / UNetMidBlock2DCrossAttn come from diffusers' unet_2d_blocks, which the reference does not
    vendor.  The stand-ins hold diffusers-named parameters and compose oracle/unet.py's LEAVES (`resnet`, `transformer2d`, `downsample`: each pinned on its own
    by tools/gen_ref_unet_leaves.py) in the published order (per layer: resnet -> attention -> append; then the downsampler -> append).  They ASSERT that the
    arguments the reference's constructor hands them (channels, temb width, head count, transformer depth, add_downsample, linear projections) agree with the
    shapes oracle.unet.init_params derives from the same configuration.

So tests/test_ref_unet_walk_cpu.py pins `oracle.unet.unet_down_mid` — embeddings, conv_in, skip order, mid block — to executed reference code; the composition INSIDE
a block and the up path (`unet_forward`'s second half) stay restated.  /root/reference is read ONLY here."""
from __future__ import annotations

import importlib
import inspect
import sys
import types
import zlib
from pathlib import Path

import torch
import torch.nn as nn

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import unet as OU  # noqa: E402
from tools import ref_shim  # noqa: E402

OUT = ROOT / "tests" / "golden"

UCFG = OU.UNetConfig(in_channels=4, out_channels=4, block_out_channels=(16, 32), layers_per_block=2, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
                     up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2), attention_head_dim=(2, 4), cross_attention_dim=24,
                     use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=8, projection_class_embeddings_input_dim=16 + 6 * 8,
                     norm_num_groups=8, norm_eps=1e-5)
SHAPES = {k: tuple(v.shape) for k, v in OU.init_params(UCFG, shapes_only=True).items()}


class _Bag(nn.Module):
    """the diffusers-named parameters under `prefix`, as one module (names keep their dots as '|')"""

    def __init__(self, prefix: str):
        super().__init__()
        self.prefix = prefix
        names = [k for k in SHAPES if k.startswith(prefix)]
        assert names, prefix
        self.p = nn.ParameterDict({k[len(prefix):].replace(".", "|"): nn.Parameter(torch.zeros(SHAPES[k])) for k in names})

    def P(self):
        return {self.prefix + k.replace("|", "."): v for k, v in self.p.items()}


_block_counter = [0]


class _DownStandIn(_Bag):
    def __init__(self, i: int, cross: bool, num_layers, transformer_layers_per_block, in_channels, out_channels, temb_channels, add_downsample, resnet_eps,
                 resnet_groups, cross_attention_dim, num_attention_heads, use_linear_projection):
        super().__init__(f"down_blocks.{i}.")
        self.i, self.has_cross_attention, self.n, self.heads, self.depth, self.down = i, cross, num_layers, num_attention_heads, transformer_layers_per_block, add_downsample
        self.groups, self.eps, self.linear = resnet_groups, resnet_eps, use_linear_projection
        pre = self.prefix
        # the reference constructor's arguments against the shapes the oracle derives from the same configuration
        assert SHAPES[pre + "resnets.0.conv1.weight"][:2] == (out_channels, in_channels) and SHAPES[pre + "resnets.0.time_emb_proj.weight"][1] == temb_channels
        assert num_layers == UCFG.layers_per_block and resnet_groups == UCFG.norm_num_groups and resnet_eps == UCFG.norm_eps
        assert add_downsample == ((pre + "downsamplers.0.conv.weight") in SHAPES)
        assert cross == UCFG.down_block_types[i].startswith("CrossAttn") == ((pre + "attentions.0.norm.weight") in SHAPES)
        if cross:
            assert num_attention_heads == UCFG.attention_head_dim[i] and transformer_layers_per_block == UCFG.transformer_layers_per_block[i]
            assert cross_attention_dim == UCFG.cross_attention_dim == SHAPES[pre + "attentions.0.transformer_blocks.0.attn2.to_k.weight"][1]
            assert use_linear_projection == UCFG.use_linear_projection and len(SHAPES[pre + "attentions.0.proj_in.weight"]) == (2 if use_linear_projection else 4)
            assert (pre + f"attentions.0.transformer_blocks.{transformer_layers_per_block - 1}.norm1.weight") in SHAPES
            assert (pre + f"attentions.0.transformer_blocks.{transformer_layers_per_block}.norm1.weight") not in SHAPES

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None, cross_attention_kwargs=None, **kw):
        assert attention_mask is None and not cross_attention_kwargs and not kw
        P, x, out = self.P(), hidden_states, ()
        for j in range(self.n):
            x = OU.resnet(P, f"{self.prefix}resnets.{j}.", x, temb, self.groups, self.eps)
            if self.has_cross_attention:
                x = OU.transformer2d(P, f"{self.prefix}attentions.{j}.", x, encoder_hidden_states, self.heads, self.depth, self.groups, self.linear)
            out += (x,)
        if self.down:
            x = OU.downsample(P, f"{self.prefix}downsamplers.0.conv", x)
            out += (x,)
        return x, out


class CrossAttnDownBlock2D(_DownStandIn):
    pass


class DownBlock2D(_DownStandIn):
    pass


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps, resnet_act_fn, transformer_layers_per_block=1,
                   num_attention_heads=None, resnet_groups=None, cross_attention_dim=None, downsample_padding=None, use_linear_projection=False,
                   only_cross_attention=False, upcast_attention=False, resnet_time_scale_shift="default", attention_head_dim=None, **kw):
    i = _block_counter[0]
    _block_counter[0] += 1
    assert resnet_act_fn == "silu" and resnet_time_scale_shift == "default" and not only_cross_attention and not upcast_attention and downsample_padding == 1 and not kw
    cls = {"CrossAttnDownBlock2D": CrossAttnDownBlock2D, "DownBlock2D": DownBlock2D}[down_block_type]
    return cls(i, down_block_type == "CrossAttnDownBlock2D", num_layers, transformer_layers_per_block, in_channels, out_channels, temb_channels, add_downsample,
               resnet_eps, resnet_groups, cross_attention_dim, num_attention_heads, use_linear_projection)


class UNetMidBlock2DCrossAttn(_Bag):
    has_cross_attention = True

    def __init__(self, transformer_layers_per_block, in_channels, temb_channels, resnet_eps, resnet_act_fn, output_scale_factor, resnet_time_scale_shift,
                 cross_attention_dim, num_attention_heads, resnet_groups, use_linear_projection, upcast_attention, **kw):
        super().__init__("mid_block.")
        assert resnet_act_fn == "silu" and output_scale_factor == 1 and resnet_time_scale_shift == "default" and not upcast_attention and not kw
        assert SHAPES["mid_block.resnets.0.conv1.weight"][:2] == (in_channels, in_channels) and SHAPES["mid_block.resnets.0.time_emb_proj.weight"][1] == temb_channels
        assert num_attention_heads == UCFG.attention_head_dim[-1] and transformer_layers_per_block == UCFG.transformer_layers_per_block[-1]
        assert cross_attention_dim == UCFG.cross_attention_dim and use_linear_projection == UCFG.use_linear_projection
        self.heads, self.depth, self.groups, self.eps, self.linear = num_attention_heads, transformer_layers_per_block, resnet_groups, resnet_eps, use_linear_projection

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None, cross_attention_kwargs=None):
        assert attention_mask is None and not cross_attention_kwargs
        P = self.P()
        x = OU.resnet(P, "mid_block.resnets.0.", hidden_states, temb, self.groups, self.eps)
        x = OU.transformer2d(P, "mid_block.attentions.0.", x, encoder_hidden_states, self.heads, self.depth, self.groups, self.linear)
        return OU.resnet(P, "mid_block.resnets.1.", x, temb, self.groups, self.eps)


class _TimestepEmbedding(ref_shim.TimestepEmbedding):
    """the lifted embedder under diffusers' call signature (act_fn = "silu", no extra condition)"""

    def __init__(self, in_channels, time_embed_dim, act_fn="silu", **kw):
        assert act_fn == "silu" and not kw
        super().__init__(in_channels, time_embed_dim)

    def forward(self, sample, condition=None):
        assert condition is None
        return super().forward(sample)


def _recording_register_to_config(init):
    """diffusers' @register_to_config: the constructor's arguments (defaults included) become `self.config`"""
    sig = inspect.signature(init)

    def wrapped(self, *a, **k):
        bound = sig.bind(self, *a, **k)
        bound.apply_defaults()
        self.register_to_config(**{n: v for n, v in bound.arguments.items() if n != "self"})
        init(self, *a, **k)

    return wrapped


class _Unbuilt:
    def __init__(self, *a, **k):
        raise AssertionError("this configuration does not construct it")


def import_controlnet():
    ref_shim.install()
    S = sys.modules
    mk = lambda name, **attrs: S.setdefault(name, types.ModuleType(name)).__dict__.update(attrs)
    mk("diffusers.loaders.single_file_model", FromOriginalModelMixin=ref_shim.FromOriginalModelMixin)
    S["diffusers.models.attention_processor"].__dict__.update(ADDED_KV_ATTENTION_PROCESSORS=(), CROSS_ATTENTION_PROCESSORS=(), AttnAddedKVProcessor=object)
    S["diffusers.models.embeddings"].__dict__.update(TextImageProjection=_Unbuilt, TextImageTimeEmbedding=_Unbuilt, TextTimeEmbedding=_Unbuilt)
    S["diffusers.utils"].__dict__.update(BaseOutput=object)
    for m in ("diffusers.models.unets", "diffusers.models.unets.unet_2d_blocks", "diffusers.models.unets.unet_2d_condition"):
        mk(m)
        S[m].__path__ = []
    S["diffusers.models.unets.unet_2d_blocks"].__dict__.update(CrossAttnDownBlock2D=CrossAttnDownBlock2D, DownBlock2D=DownBlock2D, UNetMidBlock2D=_Unbuilt,
                                                               UNetMidBlock2DCrossAttn=UNetMidBlock2DCrossAttn, get_down_block=get_down_block)
    S["diffusers.models.unets.unet_2d_condition"].__dict__.update(UNet2DConditionModel=object)
    ref_shim._pkg("simpletuner.helpers.models.kolors", ref_shim.REF / "helpers/models/kolors")
    cu, emb = S["diffusers.configuration_utils"], S["diffusers.models.embeddings"]
    keep = (cu.register_to_config, emb.TimestepEmbedding)
    cu.register_to_config, emb.TimestepEmbedding = _recording_register_to_config, _TimestepEmbedding
    try:
        return importlib.import_module("simpletuner.helpers.models.kolors.controlnet")
    finally:
        cu.register_to_config, emb.TimestepEmbedding = keep


def main():
    C = import_controlnet()
    _block_counter[0] = 0
    model = C.ControlNetModel(in_channels=UCFG.in_channels, conditioning_channels=3, flip_sin_to_cos=True, freq_shift=0, down_block_types=UCFG.down_block_types,
                              mid_block_type="UNetMidBlock2DCrossAttn", only_cross_attention=False, block_out_channels=UCFG.block_out_channels,
                              layers_per_block=UCFG.layers_per_block, downsample_padding=1, mid_block_scale_factor=1, act_fn="silu",
                              norm_num_groups=UCFG.norm_num_groups, norm_eps=UCFG.norm_eps, cross_attention_dim=UCFG.cross_attention_dim,
                              transformer_layers_per_block=UCFG.transformer_layers_per_block, attention_head_dim=UCFG.attention_head_dim,
                              use_linear_projection=UCFG.use_linear_projection, addition_embed_type=UCFG.addition_embed_type,
                              addition_time_embed_dim=UCFG.addition_time_embed_dim, projection_class_embeddings_input_dim=UCFG.projection_class_embeddings_input_dim,
                              conditioning_embedding_out_channels=(8, 16))
    model.eval()
    # seeded weights for everything the UNet shares with this class (drawn in oracle.unet.init_params' order: the oracle side rebuilds nothing, it receives P)
    seeded = OU.init_params(UCFG, seed=611)
    P = {}
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.startswith(("controlnet_down_blocks.", "controlnet_mid_block.")):      # 1x1 output convolutions -> identity: the outputs are the skip tensors
                p.copy_(torch.eye(p.shape[0]).reshape(p.shape) if p.dim() == 4 else torch.zeros_like(p))
                continue
            if name.startswith("controlnet_cond_embedding."):
                if not name.startswith("controlnet_cond_embedding.conv_out."):             # (conv_out keeps its zero initialisation: the embedding adds 0)
                    p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)) * 0.1)
                continue
            key = name.replace(".p.", ".").replace("|", ".")
            p.copy_(seeded[key])
            P[key] = p.detach().clone()
    used = {k for k in seeded if k.startswith(("time_embedding.", "add_embedding.", "conv_in.", "down_blocks.", "mid_block."))}
    assert set(P) == used, sorted(set(P) ^ used)[:8]
    g = torch.Generator().manual_seed(612)
    B, H, W, T = 2, 8, 12, 5
    inputs = {"sample": torch.randn(B, 4, H, W, generator=g), "timestep": torch.tensor([37.0, 911.0]), "encoder_hidden_states": torch.randn(B, T, 24, generator=g),
              "text_embeds": torch.randn(B, 16, generator=g), "time_ids": torch.tensor([[64.0, 96.0, 0.0, 8.0, 64.0, 96.0], [128.0, 64.0, 16.0, 0.0, 96.0, 64.0]]),
              "controlnet_cond": torch.randn(B, 3, H * 2, W * 2, generator=g)}     # (one stride-2 stage in the conditioning embedding of this configuration)
    leaves = {k: inputs[k].clone().requires_grad_(True) for k in ("sample", "encoder_hidden_states", "text_embeds")}
    for p in model.parameters():
        p.grad = None
    down, mid = model(leaves["sample"], inputs["timestep"], leaves["encoder_hidden_states"], inputs["controlnet_cond"], conditioning_scale=1.0,
                      added_cond_kwargs={"text_embeds": leaves["text_embeds"], "time_ids": inputs["time_ids"]}, return_dict=False)
    ws = [torch.randn(t.shape, generator=torch.Generator().manual_seed(620 + i)) for i, t in enumerate(list(down) + [mid])]
    sum((t * w).sum() for t, w in zip(list(down) + [mid], ws)).backward()
    grads = {}
    for name, p in model.named_parameters():
        key = name.replace(".p.", ".").replace("|", ".")
        if key in P:
            grads[key] = p.grad.detach().clone()
    cfg = {k: getattr(UCFG, k) for k in UCFG.__dataclass_fields__}
    torch.save({"config": cfg, "params": P, "inputs": inputs, "down": [t.detach().clone() for t in down], "mid": mid.detach().clone(), "w": ws, "grads": grads,
                "input_grads": {k: v.grad.detach().clone() for k, v in leaves.items()}, "registered_config": {k: v for k, v in model.config.items() if isinstance(v, (int, float, str, bool, tuple, list, type(None)))},
                "_cite": "simpletuner/helpers/models/kolors/controlnet.py:132-931 (ControlNetModel: :251 head count, :316-325 time embedding, :397-399 text_time embedder, "
                         ":431-464 down blocks, :473-486 mid block, :805-898 forward: embeddings, conv_in, down loop, mid)"}, OUT / "ref_unet_walk.pt")
    print("down:", [tuple(t.shape) for t in down], "mid:", tuple(mid.shape), "params:", len(P))


if __name__ == "__main__":
    main()
