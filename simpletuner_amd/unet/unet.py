"""UNet2DConditionModel — the MI355X-native trained component for the conv UNets (SDXL, SD1.5-style configs with head_dim 64).

Mirrors the module surface the reference drives (sdxl/model.py:350-367, unet_flowmap.py:23-44 — diffusers' UNet2DConditionModel):
`forward(sample[B,4,H,W], timestep, encoder_hidden_states[B,77,Dc], class_labels=None, added_cond_kwargs={"text_embeds","time_ids"},
return_dict=False)[0]`, diffusers state-dict key names, `.config`, `.parameters()`.  Forward and the hand-written backward are sequences of
libst355 launches:
  * activations of the conv part live as zero-bordered NHWC "grid buffers" (include/st355.h): every 3x3 / 1x1 convolution, its input
    gradient (same kernel on flipped / transposed taps) and the ResnetBlock2D additions (bias, time-embedding row, shortcut) are ONE GEMM
    launch each — the K loop walks the nine taps as row-shifted views of the input, no im2col; weight gradients are nine TN GEMMs;
  * GroupNorm(+SiLU) is a two-pass HBM kernel pair that can emit / consume dense tokens, so Transformer2DModel needs no extra layout pass
    on the way in; the way out (tokens -> grid) carries the residual add;
  * BasicTransformerBlock = affine LayerNorm kernels, fused [q|k|v] / [k|v] projections, the flash attention kernels (self: S x S;
    cross: S x 77 through the Sq != Sk entry points), GEGLU kernel, residual adds in the GEMM epilogues;
  * the two stride-2 Downsample2D convs, conv_in (4 -> 8 channels) and conv_out's input gradient go through an explicit column gather.
Weights are views of ONE bf16 arena in the native layouts (conv: [Cout, 9*Cin] = torch weight.permute(0,2,3,1)); gradients land in a second
arena of the same layout (one fused optimizer launch, contiguous slices for RCCL).  Full fine-tune only (BASELINE.json configs[1]).
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..flux.transformer import LoraGroup, _attach, _frozen
from ..ops import EPI_ADD, EPI_GEGLU, EPI_GEGLU_GRAD, EPI_HEADS, EPI_NONE
from ..training.checkpoint_plan import CheckpointPlanMixin

BF16 = torch.bfloat16
F32 = torch.float32


def _p64(t):
    r = t.shape[0]
    if r % 64 == 0 and t.is_contiguous():
        return t
    o = torch.zeros((r + 63) // 64 * 64, t.shape[1], dtype=BF16, device=t.device)
    o[:r] = t
    return o


# ST355_HEADS_FUSED=0: the attention input projections write [M, 3C] / [M, C] and st355_head_split_pad re-lays them out (A/B switch for ST355_EPI_HEADS)
_HEADS_FUSED = os.environ.get("ST355_HEADS_FUSED", "1") != "0"


class Tape:
    """reverse-mode record of the step: (outputs, inputs, backward closure); gradients that meet on one tensor are summed with ops.add"""

    def __init__(self):
        self.ops: List[tuple] = []

    def rec(self, outs, ins, fn, acc: Optional[int] = None):
        """acc = index of an input whose backward closure ADDS INTO the gradient already accumulated on that tensor: `fn(*douts, dres=<that gradient or None>)`
        returns the sum for it (the LayerNorm backward kernel's residual operand: the sum rides in its own pass instead of a separate ops.add launch —
        210 adds per SDXL step, 7 ms at batch 16: rocprofv3 r06)"""
        self.ops.append((outs, ins, fn, acc))

    def backward(self, out, dout, wrt=None):
        """sweep the record backwards from d(out) = dout; `wrt`: tensors whose accumulated gradients are returned (a checkpointed unit's inputs)"""
        grads = {id(out): dout}
        while self.ops:
            outs, ins, fn, acc = self.ops.pop()     # popping frees the closure's saved activations as the sweep proceeds
            douts = [grads.pop(id(o), None) for o in outs]
            if all(d is None for d in douts):
                continue
            if acc is not None and ins[acc] is not None:
                dins = fn(*douts, dres=grads.pop(id(ins[acc]), None))
            else:
                dins = fn(*douts)
            for i, d in zip(ins, dins):
                if i is None or d is None:
                    continue
                k = id(i)
                grads[k] = ops.add(grads[k], d) if k in grads else d
        return None if wrt is None else [grads.get(id(t)) for t in wrt]


class UNet2DConditionModel(CheckpointPlanMixin, nn.Module):
    def __init__(self, in_channels: int = 4, out_channels: int = 4, block_out_channels=(320, 640, 1280), layers_per_block: int = 2,
                 down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                 up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2, 10),
                 attention_head_dim=(5, 10, 20), cross_attention_dim: int = 2048, use_linear_projection: bool = True,
                 addition_embed_type: Optional[str] = "text_time", addition_time_embed_dim: int = 256,
                 projection_class_embeddings_input_dim: int = 2816, norm_num_groups: int = 32, norm_eps: float = 1e-5, sample_size: int = 128,
                 device=None, **_ignored):
        super().__init__()
        nb = len(block_out_channels)
        if isinstance(transformer_layers_per_block, int):
            transformer_layers_per_block = (transformer_layers_per_block,) * nb
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * nb
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                                      transformer_layers_per_block=tuple(transformer_layers_per_block), attention_head_dim=tuple(attention_head_dim),
                                      cross_attention_dim=cross_attention_dim, use_linear_projection=use_linear_projection,
                                      addition_embed_type=addition_embed_type, addition_time_embed_dim=addition_time_embed_dim,
                                      projection_class_embeddings_input_dim=projection_class_embeddings_input_dim, norm_num_groups=norm_num_groups,
                                      norm_eps=norm_eps, sample_size=sample_size)
        c = self.config
        if in_channels > 8 or out_channels > 8:
            raise ValueError("conv_in / conv_out are built for <= 8 latent channels")
        for ch, nh, typ in zip(block_out_channels, attention_head_dim, down_block_types):
            if ch % 64 or ch % norm_num_groups:
                raise ValueError("block_out_channels must be multiples of 64 (GEMM K granule) and of the group count")
            if typ.startswith("CrossAttn") and (ch % nh or (ch // nh) % 8):
                raise ValueError(f"attention head_dim {ch / nh} must be a multiple of 8")
        if cross_attention_dim % 64 or (addition_embed_type not in (None, "text_time")):
            raise ValueError("unsupported cross_attention_dim / addition_embed_type")
        self.device_ = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._specs: List[SimpleNamespace] = []
        self._build()
        total = 0
        for s in self._specs:
            s.off = total
            total += (s.numel + 7) // 8 * 8
        self.arena = torch.zeros(total, dtype=BF16, device=self.device_)
        for s in self._specs:
            s.t = self.arena[s.off:s.off + s.numel].view(*s.shape)
            s.g = None
            if s.views is None:
                _attach(self, s.name, _frozen(s.t))
            else:                                   # fused projections: the diffusers keys are row-slices of the one matrix
                for nm, lo, hi in s.views:
                    _attach(self, nm, _frozen(s.t[lo:hi]))
        self.grad_arena = None
        self.full = False
        self.grad_sync = None
        self._last_grad_flat = None
        self._full_params: List[nn.Parameter] = []
        self._tmp: Dict = {}
        self.lora_groups: List[LoraGroup] = []
        self.lora_flat = self.lora_grad_flat = None
        self._lora_params: List[nn.Parameter] = []
        self._probe = None

    # ------------------------------------------------------------------------------------------------
    # construction: parameter slots in arena order + the layer graph
    # ------------------------------------------------------------------------------------------------
    def _slot(self, name, *shape, kind="w", views=None):
        n = 1
        for d in shape:
            n *= d
        s = SimpleNamespace(name=name, shape=shape, numel=n, kind=kind, t=None, g=None, off=0, views=views)
        self._specs.append(s)
        return s

    def _lin(self, name, out_f, in_f, bias=True):
        return SimpleNamespace(kind="lin", name=name, w=self._slot(name + ".weight", out_f, in_f), b=self._slot(name + ".bias", out_f, kind="b") if bias else None,
                               wT=None, N=out_f, K=in_f, lora=None, is_geglu_proj=name.endswith("ff.net.0.proj"), w_il=None, b_il=None, wT_il=None)

    def _lin_fused(self, prefix, names, out_each, in_f):
        """projections that share an input stored as one matrix; the per-projection diffusers keys are row-slices (registered as views later)"""
        views = [(f"{prefix}{nm}.weight", j * out_each, (j + 1) * out_each) for j, nm in enumerate(names)]
        l = SimpleNamespace(kind="lin", name=prefix + "+".join(names), w=self._slot(prefix + "+".join(names) + ".weight", len(names) * out_each, in_f, views=views),
                            b=None, wT=None, N=len(names) * out_each, K=in_f, fused=(prefix, names, out_each), lora=None, is_geglu_proj=False, w_il=None, b_il=None,
                            wT_il=None)
        return l

    def _conv(self, name, cin, cout, taps=9, cout_pad=None, cin_cols=None):
        co = cout_pad or cout
        cols = cin_cols or taps * cin
        return SimpleNamespace(kind="conv", name=name, w=self._slot(name + ".weight", co, cols), b=self._slot(name + ".bias", co, kind="b"), wT=None,
                               cin=cin, cout=co, cout_true=cout, taps=taps)

    def _norm(self, name, ch):
        return SimpleNamespace(w=self._slot(name + ".weight", ch, kind="g"), b=self._slot(name + ".bias", ch, kind="b"))

    def _resnet(self, p, cin, cout, temb):
        return SimpleNamespace(norm1=self._norm(p + "norm1", cin), conv1=self._conv(p + "conv1", cin, cout), temb=self._lin(p + "time_emb_proj", cout, temb),
                               norm2=self._norm(p + "norm2", cout), conv2=self._conv(p + "conv2", cout, cout),
                               short=self._conv(p + "conv_shortcut", cin, cout, taps=1) if cin != cout else None, cin=cin, cout=cout)

    def _transformer(self, p, ch, heads, n_layers):
        c = self.config
        tr = SimpleNamespace(C=ch, heads=heads, norm=self._norm(p + "norm", ch), proj_in=self._lin(p + "proj_in", ch, ch), blocks=[])
        for k in range(n_layers):
            q = f"{p}transformer_blocks.{k}."
            tr.blocks.append(SimpleNamespace(
                norm1=self._norm(q + "norm1", ch), qkv=self._lin_fused(q + "attn1.", ["to_q", "to_k", "to_v"], ch, ch), out1=self._lin(q + "attn1.to_out.0", ch, ch),
                norm2=self._norm(q + "norm2", ch), q2=self._lin(q + "attn2.to_q", ch, ch, bias=False),
                kv2=self._lin_fused(q + "attn2.", ["to_k", "to_v"], ch, c.cross_attention_dim), out2=self._lin(q + "attn2.to_out.0", ch, ch),
                norm3=self._norm(q + "norm3", ch), ff1=self._lin(q + "ff.net.0.proj", 8 * ch, ch), ff2=self._lin(q + "ff.net.2", ch, 4 * ch)))
        tr.proj_out = self._lin(p + "proj_out", ch, ch)
        return tr

    def _build(self):
        c = self.config
        ch = c.block_out_channels
        nb = len(ch)
        temb = 4 * ch[0]
        self.temb_dim = temb
        self.l_conv_in = self._conv("conv_in", 8, ch[0], taps=1, cin_cols=128)            # columns: tap*8 + c (c < in_channels), zero-padded 72 -> 128
        self.t1 = self._lin("time_embedding.linear_1", temb, ch[0])
        self.t2 = self._lin("time_embedding.linear_2", temb, temb)
        if c.addition_embed_type == "text_time":
            self.a1 = self._lin("add_embedding.linear_1", temb, c.projection_class_embeddings_input_dim)
            self.a2 = self._lin("add_embedding.linear_2", temb, temb)
        self.down = []
        cin = ch[0]
        skip = [ch[0]]
        for i, typ in enumerate(c.down_block_types):
            blk = SimpleNamespace(resnets=[], attns=[], down=None)
            for j in range(c.layers_per_block):
                blk.resnets.append(self._resnet(f"down_blocks.{i}.resnets.{j}.", cin, ch[i], temb))
                cin = ch[i]
                if typ.startswith("CrossAttn"):
                    blk.attns.append(self._transformer(f"down_blocks.{i}.attentions.{j}.", cin, c.attention_head_dim[i], c.transformer_layers_per_block[i]))
                skip.append(cin)
            if i < nb - 1:
                blk.down = self._conv(f"down_blocks.{i}.downsamplers.0.conv", cin, cin)
                skip.append(cin)
            self.down.append(blk)
        self.mid = SimpleNamespace(r0=self._resnet("mid_block.resnets.0.", cin, cin, temb),
                                   attn=self._transformer("mid_block.attentions.0.", cin, c.attention_head_dim[-1], c.transformer_layers_per_block[-1]),
                                   r1=self._resnet("mid_block.resnets.1.", cin, cin, temb))
        self.up = []
        for i, typ in enumerate(c.up_block_types):
            ri = nb - 1 - i
            blk = SimpleNamespace(resnets=[], attns=[], up=None)
            for j in range(c.layers_per_block + 1):
                blk.resnets.append(self._resnet(f"up_blocks.{i}.resnets.{j}.", cin + skip.pop(), ch[ri], temb))
                cin = ch[ri]
                if typ.startswith("CrossAttn"):
                    blk.attns.append(self._transformer(f"up_blocks.{i}.attentions.{j}.", cin, c.attention_head_dim[ri], c.transformer_layers_per_block[ri]))
            if i < nb - 1:
                blk.up = self._conv(f"up_blocks.{i}.upsamplers.0.conv", cin, cin)
            self.up.append(blk)
        self.norm_out = self._norm("conv_norm_out", cin)
        self.l_conv_out = self._conv("conv_out", cin, c.out_channels, cout_pad=8)

    # ------------------------------------------------------------------------------------------------
    # weights: synthetic init, diffusers <-> native layout
    # ------------------------------------------------------------------------------------------------
    def _convs(self):
        return [s for s in self._all_layers() if s.kind == "conv"]

    def _all_layers(self):
        out = [self.l_conv_in, self.t1, self.t2]
        if self.config.addition_embed_type == "text_time":
            out += [self.a1, self.a2]

        def res(r):
            return [r.conv1, r.temb, r.conv2] + ([r.short] if r.short is not None else [])

        def tr(t):
            o = [t.proj_in]
            for b in t.blocks:
                o += [b.qkv, b.out1, b.q2, b.kv2, b.out2, b.ff1, b.ff2]
            return o + [t.proj_out]

        for blk in self.down:
            for j, r in enumerate(blk.resnets):
                out += res(r)
                if blk.attns:
                    out += tr(blk.attns[j])
            if blk.down is not None:
                out.append(blk.down)
        out += res(self.mid.r0) + tr(self.mid.attn) + res(self.mid.r1)
        for blk in self.up:
            for j, r in enumerate(blk.resnets):
                out += res(r)
                if blk.attns:
                    out += tr(blk.attns[j])
            if blk.up is not None:
                out.append(blk.up)
        out.append(self.l_conv_out)
        return out

    @torch.no_grad()
    def init_synthetic(self, seed: int = 42):
        g = torch.Generator(device=self.device_).manual_seed(seed)
        for s in self._specs:
            if s.kind == "g":
                s.t.copy_(1.0 + 0.1 * torch.randn(s.shape, generator=g, device=self.device_))
            elif s.kind == "b":
                s.t.copy_(0.02 * torch.randn(s.shape, generator=g, device=self.device_))
            else:
                s.t.copy_(torch.randn(s.shape, generator=g, device=self.device_, dtype=F32) * (1.0 / math.sqrt(s.shape[-1])))
        self._zero_padding()

    @torch.no_grad()
    def _zero_padding(self):
        ci = self.config.in_channels
        w = self.l_conv_in.w.t
        w[:, 72:] = 0
        w[:, :72].view(w.shape[0], 9, 8)[:, :, ci:] = 0
        self.l_conv_out.w.t[self.config.out_channels:] = 0
        self.l_conv_out.b.t[self.config.out_channels:] = 0

    @torch.no_grad()
    def diffusers_state_dict(self) -> Dict[str, torch.Tensor]:
        """weights in diffusers' key names and shapes (conv [O,I,kh,kw]; fused projections split)"""
        sd = {}
        c = self.config
        for l in self._all_layers():
            if l.kind == "conv":
                w = l.w.t
                if l is self.l_conv_in:
                    sd[l.name + ".weight"] = w[:, :72].reshape(-1, 3, 3, 8)[..., :c.in_channels].permute(0, 3, 1, 2).contiguous()
                    sd[l.name + ".bias"] = l.b.t.clone()
                    continue
                k = 3 if l.taps == 9 else 1
                w4 = w.view(l.cout, k, k, l.cin).permute(0, 3, 1, 2)[:l.cout_true].contiguous()
                sd[l.name + ".weight"], sd[l.name + ".bias"] = w4, l.b.t[:l.cout_true].clone()
            elif hasattr(l, "fused"):
                prefix, names, each = l.fused
                for j, nm in enumerate(names):
                    sd[f"{prefix}{nm}.weight"] = l.w.t[j * each:(j + 1) * each].clone()
            else:
                sd[l.name + ".weight"] = l.w.t.clone()
                if l.b is not None:
                    sd[l.name + ".bias"] = l.b.t.clone()
        for s in self._specs:
            if s.kind == "g":
                sd[s.name] = s.t.clone()
                sd[s.name[:-len("weight")] + "bias"] = dict((q.name, q) for q in self._specs)[s.name[:-len("weight")] + "bias"].t.clone()
        if not c.use_linear_projection:
            for k in list(sd):
                if k.endswith("proj_in.weight") or k.endswith("proj_out.weight"):
                    sd[k] = sd[k][:, :, None, None]
        return sd

    @torch.no_grad()
    def load_diffusers_state(self, sd: Dict[str, torch.Tensor]):
        c = self.config
        for l in self._all_layers():
            if l.kind == "conv":
                w4 = sd[l.name + ".weight"].to(self.device_, BF16)
                if l is self.l_conv_in:
                    l.w.t.zero_()
                    l.w.t[:, :72].view(-1, 9, 8)[:, :, :c.in_channels] = w4.permute(0, 2, 3, 1).reshape(w4.shape[0], 9, c.in_channels)
                else:
                    l.w.t.zero_()
                    l.w.t[:l.cout_true] = w4.permute(0, 2, 3, 1).reshape(l.cout_true, -1)
                l.b.t.zero_()
                l.b.t[:l.cout_true] = sd[l.name + ".bias"].to(self.device_, BF16)
            elif hasattr(l, "fused"):
                prefix, names, each = l.fused
                for j, nm in enumerate(names):
                    l.w.t[j * each:(j + 1) * each] = sd[f"{prefix}{nm}.weight"].to(self.device_, BF16)
            else:
                l.w.t.copy_(sd[l.name + ".weight"].to(self.device_, BF16).reshape(l.w.shape))
                if l.b is not None:
                    l.b.t.copy_(sd[l.name + ".bias"].to(self.device_, BF16))
        for s in self._specs:
            if s.kind == "g":
                s.t.copy_(sd[s.name].to(self.device_, BF16))
                bn = s.name[:-len("weight")] + "bias"
                dict((q.name, q) for q in self._specs)[bn].t.copy_(sd[bn].to(self.device_, BF16))

    # ------------------------------------------------------------------------------------------------
    # full fine-tune plumbing
    # ------------------------------------------------------------------------------------------------
    def enable_full_finetune(self):
        self.full = True
        self.grad_arena = torch.zeros_like(self.arena)
        for s in self._specs:
            s.g = self.grad_arena[s.off:s.off + s.numel].view(*s.shape)
        ps = sorted([p for _, p in self.named_parameters()], key=lambda p: p.data_ptr())
        for p in ps:
            p.requires_grad_(True)
        self._full_params = ps
        base = self.arena.data_ptr()
        self._full_offsets = [((p.data_ptr() - base) // 2, p.numel()) for p in ps]
        self._alloc_transposed()
        return ps

    def _alloc_transposed(self):
        dev = self.device_
        for l in self._all_layers():
            if l.wT is not None:
                continue
            if l.kind == "conv" and l is not self.l_conv_in:
                if l.taps == 9 and l is not self.l_conv_out and not self._is_down(l):
                    l.wT = torch.empty(l.cin, 9 * l.cout, dtype=BF16, device=dev)          # flipped taps, transposed: the dgrad conv's weight
                elif l is self.l_conv_out:
                    l.wT = torch.zeros(l.cin, 128, dtype=BF16, device=dev)                 # [ci, tap'*8 + co], columns 72.. stay zero
                else:
                    l.wT = torch.empty(l.w.shape[1], l.cout, dtype=BF16, device=dev)       # plain transpose (1x1 shortcut, stride-2 columns)
            elif l.kind == "lin":
                l.wT = torch.empty(l.K, l.N, dtype=BF16, device=dev)

    # ------------------------------------------------------------------------------------------------
    # LoRA (peft naming) on the attention projections: DEFAULT_LORA_TARGET to_k,to_q,to_v,to_out.0 (sdxl/model.py, common.py:1049-1128)
    # ------------------------------------------------------------------------------------------------
    def add_lora_adapter(self, rank: int = 16, alpha: Optional[float] = None, seed: int = 7, init_b_std: float = 0.0):
        if self.full:
            raise RuntimeError("full fine-tune and LoRA adapters are exclusive")
        alpha = float(rank if alpha is None else alpha)
        dev = self.device_
        plan = []
        self.lora_groups = []

        def group(lin, prefix, names):
            each = lin.N // len(names)
            g = LoraGroup(lin.K, lin.N, [(prefix + n, j * each, each) for j, n in enumerate(names)], rank, alpha, dev)
            lin.lora = g
            self.lora_groups.append(g)
            for (name, _, N) in g.targets:
                plan.append((g, name, N, lin.K))

        def tr(t, p):
            for k, b in enumerate(t.blocks):
                q = f"{p}transformer_blocks.{k}."
                group(b.qkv, q + "attn1.", ["to_q", "to_k", "to_v"]); group(b.out1, q + "attn1.", ["to_out.0"])
                group(b.q2, q + "attn2.", ["to_q"]); group(b.kv2, q + "attn2.", ["to_k", "to_v"]); group(b.out2, q + "attn2.", ["to_out.0"])

        for i, blk in enumerate(self.down):
            for j, t in enumerate(blk.attns):
                tr(t, f"down_blocks.{i}.attentions.{j}.")
        tr(self.mid.attn, "mid_block.attentions.0.")
        for i, blk in enumerate(self.up):
            for j, t in enumerate(blk.attns):
                tr(t, f"up_blocks.{i}.attentions.{j}.")
        total = (sum(rank * K + N * rank for (_, _, N, K) in plan) + 7) // 8 * 8
        self.lora_flat = torch.zeros(total, dtype=F32, device=dev)
        self.lora_grad_flat = torch.zeros(total, dtype=F32, device=dev)
        gen = torch.Generator(device=dev).manual_seed(seed)
        off = 0
        self._lora_params = []
        for (g, name, N, K) in plan:
            if not g.A:
                g.flat_lo = off
            a = self.lora_flat[off:off + rank * K].view(rank, K); ga = self.lora_grad_flat[off:off + rank * K].view(rank, K)
            off += rank * K
            b = self.lora_flat[off:off + N * rank].view(N, rank); gb = self.lora_grad_flat[off:off + N * rank].view(N, rank)
            off += N * rank
            a.copy_((torch.rand(rank, K, generator=gen, device=dev) * 2 - 1) / math.sqrt(K))
            if init_b_std > 0:
                b.copy_(torch.randn(N, rank, generator=gen, device=dev) * init_b_std)
            pa, pb = nn.Parameter(a), nn.Parameter(b)
            _attach(self, name + ".lora_A.default.weight", pa); _attach(self, name + ".lora_B.default.weight", pb)
            g.A.append(pa.data); g.B.append(pb.data); g.gA.append(ga); g.gB.append(gb)
            g.flat_hi = off
            self._lora_params += [pa, pb]
        self._alloc_transposed()
        self._refresh_transposed()                       # frozen base: the K-major copies are built once
        return self._lora_params

    def _is_down(self, l):
        return any(blk.down is l for blk in self.down)

    def trainable_parameters(self):
        return list(self._full_params) if self.full else list(self._lora_params)

    @torch.no_grad()
    def _refresh_transposed(self):
        for l in self._all_layers():
            if l.wT is None:
                continue
            if l.kind == "conv" and l.taps == 9 and l is not self.l_conv_out and not self._is_down(l):
                l.wT.view(l.cin, 9, l.cout).copy_(l.w.t.view(l.cout, 9, l.cin).flip(1).permute(2, 1, 0))
            elif l is self.l_conv_out:
                l.wT[:, :72].view(l.cin, 9, 8).copy_(l.w.t.view(8, 9, l.cin).flip(1).permute(2, 1, 0))
            else:
                ops.transpose(l.w.t, out=l.wT)
        if not self.full:
            self._prepare_geglu()                       # the interleaved feed-forward copies follow the (frozen) weights

    def _ready(self, *slots):
        """gradient slices that are final: hand them to the bucketed all-reduce (overlaps the rest of the backward); adjacent slices merge"""
        gs = self.grad_sync
        if gs is None or not self.full:
            return
        for sl in slots:
            if sl is not None:
                gs.ready(sl.off, sl.off + (sl.numel + 7) // 8 * 8)

    def _f32(self, *shape):
        t = self._tmp.get(shape)
        if t is None:
            t = self._tmp[shape] = torch.empty(*shape, dtype=F32, device=self.device_)
        return t

    def _bias_grad(self, dy2d, slot, rows=None):
        t = self._f32(1, dy2d.shape[1])
        ops.colsum_prod(dy2d if rows is None else dy2d[:rows], t)
        slot.g.copy_(t[0])

    # ------------------------------------------------------------------------------------------------
    # ops with their backward closures
    # ------------------------------------------------------------------------------------------------
    def _linear(self, T, l, x, residual=None, need_dx=True):
        lo = l.lora
        kw = {}
        Tl = None
        if lo is not None:                                  # y = x W^T + (s B)(A x): the low-rank term rides the GEMM's K-extension
            Tl = ops.gemm(x, lo.A_cat)
            kw = dict(a2=Tl, b2=lo.B_blk)
        y = ops.gemm(x, l.w.t, bias=None if l.b is None else l.b.t, epilogue=EPI_ADD if residual is not None else EPI_NONE, aux_in=residual, **kw)
        if T is not None:
            def bwd(dy):
                return self._linear_bwd(l, x, Tl, dy, need_dx, y=y), (dy if residual is not None else None)
            T.rec([y], [x if need_dx else None, residual], bwd)
        return y

    def _linear_bwd(self, l, x, Tl, dy, need_dx=True, y=None):
        """weight / bias / adapter gradients of y = x W^T (+ low-rank term) from dy, and dx = dy W (+ rank-space term)"""
        lo = l.lora
        if self.full:
            ops.gemm_tn(_p64(dy), _p64(x), out=l.w.g)
            if l.b is not None:
                self._bias_grad(dy, l.b)
            self._ready(l.b, l.w) if (l.b is not None and l.b.off > l.w.off) else self._ready(l.w, l.b)
        kb = {}
        if lo is not None:
            U = ops.gemm(dy, lo.B_blk_T)
            lo.grads(x, Tl, dy, U, False, None)
            kb = dict(a2=U, b2=lo.A_cat_T)
        if self._probe is not None:               # lab hook (tools/sdxl_lora_outlier_probe.py): the operands of this layer's backward, by reference
            self._probe("linear", l.name, dict(x=x, dy=dy, y=y))
        return ops.gemm(dy, l.wT, **kb) if need_dx else None

    def _conv3(self, T, l, x, B, H, W, img_add=None, residual=None):
        y = ops.conv(x, l.w.t, B, H, W, bias=l.b.t, img_add=img_add, residual=residual, taps=l.taps)
        if T is not None:
            n = B * (H + 2) * (W + 2)

            def bwd(dy):
                if self.full:
                    ops.conv_wgrad(x, dy, l.w.g, B, H, W, taps=l.taps)
                    self._bias_grad(dy, l.b, rows=n)
                    self._ready(l.b, l.w)
                dx = ops.conv(dy, l.wT, B, H, W, taps=l.taps)
                dadd = None
                if img_add is not None:
                    t = self._f32(B, l.cout)
                    ops.colsum_prod(dy[:n], t, rows_per_batch=(H + 2) * (W + 2))
                    dadd = t.to(BF16)
                return dx, dadd, (dy if residual is not None else None)
            T.rec([y], [x, img_add, residual], bwd)
        return y

    def _gn(self, T, nm, x, B, H, W, eps, silu=True, tokens=False):
        y, stats = ops.groupnorm_fwd(x, nm.w.t, nm.b.t, B, H, W, groups=self.config.norm_num_groups, eps=eps, silu=silu, out_tokens=tokens)
        if T is not None:
            def bwd(dy):
                Cn = x.shape[1]
                dg, db = (self._f32(Cn), self._f32(Cn, 1).view(Cn)) if self.full else (None, None)
                dx = ops.groupnorm_bwd(dy, x, nm.w.t, nm.b.t, stats, B, H, W, groups=self.config.norm_num_groups, silu=silu, dy_tokens=tokens, dgamma=dg,
                                       dbeta=db)
                if self.full:
                    nm.w.g.copy_(dg); nm.b.g.copy_(db)
                    self._ready(nm.b, nm.w)
                return (dx,)
            T.rec([y], [x], bwd)
        return y

    def _ln(self, T, nm, h):
        n = ops.layernorm_fwd(h, nm.w.t, nm.b.t, eps=1e-5)
        if T is not None:
            def bwd(dn, dres=None):
                if self.full:
                    D = h.shape[1]
                    dw, db = self._f32(D), self._f32(D, 1)
                    ops.layernorm_param_grads(dn, h, dw, db.view(D), eps=1e-5)
                    nm.w.g.copy_(dw); nm.b.g.copy_(db.view(D))
                    self._ready(nm.b, nm.w)
                return (ops.layernorm_bwd(dn, h, nm.w.t, dres=dres, eps=1e-5),)       # + the gradient the residual stream already carries (Tape.rec acc)
            T.rec([n], [h], bwd, acc=0)
        return n

    def _resnet_fwd(self, T, r, x, se, B, H, W):
        eps = self.config.norm_eps
        h = self._gn(T, r.norm1, x, B, H, W, eps)
        tp = self._linear(T, r.temb, se)
        h = self._conv3(T, r.conv1, h, B, H, W, img_add=tp)
        h = self._gn(T, r.norm2, h, B, H, W, eps)
        sc = x if r.short is None else self._conv3(T, r.short, x, B, H, W)
        return self._conv3(T, r.conv2, h, B, H, W, residual=sc)

    def _self_attn(self, T, qkv, B, S, heads):
        C_ = qkv.shape[1] // 3
        return self._attention(T, qkv, qkv, 0, C_, 2 * C_, C_, B, S, S, heads)

    def _cross_attn(self, T, q, kv, B, S, Sk, heads):
        C_ = q.shape[1]
        return self._attention(T, q, kv, 0, 0, C_, C_, B, S, Sk, heads)

    def _attention(self, T, qsrc, kvsrc, qo, ko, vo, C_, B, S, Sk, heads):
        """softmax(q k^T / sqrt(hd)) v over `heads` heads of width hd = C_/heads; q / k / v are column blocks (offsets qo / ko / vo) of the token-major
        projection outputs.  hd in {64, 128}: the flash kernels on the buffers as they are; other hd <= 128 (SD1.5's 40 -> 64, 80 -> 96; 81..128 -> 128): zero-padded
        heads; hd > 128 (SD1.5's 160, at <= 256 tokens): unfused GEMM - softmax - GEMM per head."""
        hd = C_ // heads
        hp = 64 if hd <= 64 else 96 if hd <= 80 else 128 if hd <= 128 else 0      # (the head_dim-96 kernels contract over 80 channels: a head wider than 80 pads to 128)
        if hp == 0:
            return self._attention_unfused(T, qsrc, kvsrc, qo, ko, vo, C_, B, S, Sk, heads)
        scale = 1.0 / math.sqrt(hd)
        dev = qsrc.device
        self_attn = qsrc is kvsrc
        M, Mk = B * S, B * Sk
        Q, Qt, Sp = ops.head_split(qsrc[:, qo:qo + C_], B, heads, hp, S, d_src=hd, want_xt=not ops.ATTN_TR)
        K, Kt, Skp = ops.head_split(kvsrc[:, ko:ko + C_], B, heads, hp, Sk, d_src=hd, want_xt=not ops.ATTN_TR)
        _, Vt, _ = ops.head_split(kvsrc[:, vo:vo + C_], B, heads, hp, Sk, want_x=False, d_src=hd)
        exact = hp == hd
        Cp = heads * hp
        Op = torch.empty(M, Cp, dtype=BF16, device=dev)
        lse = torch.empty(B, heads, S, dtype=F32, device=dev)
        # training: the forward also keeps the rounding residual of O, so that the backward's delta = rowsum(dO * O) is taken from the un-rounded output.  The UNet's
        # attention inputs are LayerNorm outputs with a large component common to all tokens (0.97 of the row norm at SDXL's 32^2 level); a flash-style backward that
        # reads delta from the bf16 O leaves that component un-cancelled in dQ / dK (rel-L2 0.26 on dQ there: profiles/r05_sdxl_lora_outlier_probe.log)
        # Self-attention only: the common component comes from the LayerNorm output feeding BOTH q and k; over the 77 text keys of attn2 there is no such common key
        # component to cancel, and the residual cost every cross-attention an O-sized write + read with no measured benefit (r5 advice)
        Ores = torch.empty_like(Op) if (T is not None and self_attn) else None
        if self_attn:
            ops.attn_fwd(Q, K, Vt, Op, lse, B, heads, S, Sp, hp, scale, O_res=Ores)
        else:
            ops.attn_cross_fwd(Q, K, Vt, Op, lse, B, heads, S, Sk, Skp, hp, scale, O_res=Ores)
        O = Op if exact else Op.view(M, heads, hp)[:, :, :hd].reshape(M, C_)

        def pad_cols(t2d, rows):                      # [rows, heads*hd] (any row stride) -> [rows, heads*hp] zero-padded heads
            o = torch.zeros(rows, heads, hp, dtype=BF16, device=dev)
            o[:, :, :hd] = t2d.reshape(rows, heads, hd)
            return o.view(rows, Cp)

        if T is not None:
            def bwd(dO):
                dq_src = torch.empty_like(qsrc)
                dkv_src = dq_src if self_attn else torch.empty_like(kvsrc)
                dQ, dK = torch.empty_like(Q), torch.empty_like(K)
                if exact:
                    v_rows, dv_rows, dOp = kvsrc[:, vo:vo + C_], dkv_src[:, vo:vo + C_], dO
                else:
                    v_rows, dOp = pad_cols(kvsrc[:, vo:vo + C_], Mk), pad_cols(dO, M)
                    dv_rows = torch.empty(Mk, Cp, dtype=BF16, device=dev)
                if self_attn:
                    ops.attn_bwd(Q, K, Qt, Kt, v_rows, Op, dOp, lse, dQ, dK, dv_rows, B, heads, S, Sp, hp, scale, O_res=Ores)
                else:
                    ops.attn_cross_bwd(Q, K, Qt, Kt, v_rows, Op, dOp, lse, dQ, dK, dv_rows, B, heads, S, Sp, Sk, Skp, hp, scale, O_res=Ores)
                if not exact:
                    dkv_src[:, vo:vo + C_] = dv_rows.view(Mk, heads, hp)[:, :, :hd].reshape(Mk, C_)
                ops.head_merge(dQ, dq_src[:, qo:qo + C_], B, heads, hp, S, d_src=hd)
                ops.head_merge(dK, dkv_src[:, ko:ko + C_], B, heads, hp, Sk, d_src=hd)
                if self._probe is not None:
                    self._probe("attn", None, dict(qsrc=qsrc, kvsrc=kvsrc, O=O, dO=dO, dq_src=dq_src, dkv_src=dkv_src, heads=heads, B=B, S=S, Sk=Sk, self_attn=self_attn))
                return (dq_src,) if self_attn else (dq_src, dkv_src)
            T.rec([O], [qsrc] if self_attn else [qsrc, kvsrc], bwd)
        return O

    def _heads_fused_ok(self, C_, heads, B, S):
        """the head-splitting projection epilogue (ST355_EPI_HEADS) applies: 64-wide heads, the transposing-read backward (no Q^T / K^T copies), 8-token V^T granules"""
        return _HEADS_FUSED and ops.ATTN_TR and self._probe is None and C_ == heads * 64 and S % 8 == 0 and B * S >= 256

    def _lora_fwd(self, l, x):
        lo = l.lora
        if lo is None:
            return None, {}
        Tl = ops.gemm(x, lo.A_cat)
        return Tl, dict(a2=Tl, b2=lo.B_blk)

    def _self_attn_proj_fused(self, T, l, n1, B, S, heads):
        """attn1 of a BasicTransformerBlock (to_q | to_k | to_v as ONE projection, then softmax(q k^T / 8) v) with the head split inside the projection GEMM's
        epilogue (ST355_EPI_HEADS): q / k leave head-major, v row-major + head-major V^T — the three st355_head_split_pad passes over the [M, 3C] projection of the
        unfused form are gone, and [M, 3C] itself is never stored.  Same accumulators, bias add and bf16 rounding: bit-identical attention operands."""
        C_ = l.N // 3
        M = B * S
        dev = n1.device
        Sp = (S + 63) // 64 * 64
        Q = torch.empty(B, heads, S, 64, dtype=BF16, device=dev)
        K = torch.empty(B, heads, S, 64, dtype=BF16, device=dev)
        Vt = torch.empty(B, heads, 64, Sp, dtype=BF16, device=dev)
        if Sp > S:
            Vt[..., S:].zero_()
        v_rows = torch.empty(M, C_, dtype=BF16, device=dev)
        Tl, kw = self._lora_fwd(l, n1)
        ops.gemm(n1, l.w.t, bias=None if l.b is None else l.b.t, out=v_rows, epilogue=EPI_HEADS, heads=ops.heads(Q, K, Vt, heads, S, 0, C_, C_), rows_per_batch=S, **kw)
        scale = 0.125
        Op = torch.empty(M, C_, dtype=BF16, device=dev)
        lse = torch.empty(B, heads, S, dtype=F32, device=dev)
        Ores = torch.empty_like(Op) if T is not None else None          # (see _attention: delta from the un-rounded output)
        ops.attn_fwd(Q, K, Vt, Op, lse, B, heads, S, Sp, 64, scale, O_res=Ores)
        if T is not None:
            def bwd(dO):
                dqkv = torch.empty(M, 3 * C_, dtype=BF16, device=dev)
                dQ, dK = torch.empty_like(Q), torch.empty_like(K)
                ops.attn_bwd(Q, K, None, None, v_rows, Op, dO, lse, dQ, dK, dqkv[:, 2 * C_:], B, heads, S, Sp, 64, scale, O_res=Ores)
                ops.head_merge(dQ, dqkv[:, :C_], B, heads, 64, S, d_src=64)
                ops.head_merge(dK, dqkv[:, C_:2 * C_], B, heads, 64, S, d_src=64)
                return (self._linear_bwd(l, n1, Tl, dqkv),)
            T.rec([Op], [n1], bwd)
        return Op

    def _cross_attn_proj_fused(self, T, lq, n2, kv, B, S, Sk, heads):
        """attn2: the query projection writes head-major q from its epilogue (ST355_EPI_HEADS, q part only); the 77 text keys keep the head-split kernel"""
        C_ = lq.N
        M, Mk = B * S, B * Sk
        dev = n2.device
        Sp = (S + 63) // 64 * 64
        Q = torch.empty(B, heads, S, 64, dtype=BF16, device=dev)
        Tl, kw = self._lora_fwd(lq, n2)
        ops.gemm(n2, lq.w.t, bias=None if lq.b is None else lq.b.t, epilogue=EPI_HEADS, heads=ops.heads(Q, None, None, heads, S, 0, C_, 0), rows_per_batch=S, **kw)
        K, _, Skp = ops.head_split(kv[:, :C_], B, heads, 64, Sk, d_src=64, want_xt=False)
        _, Vt, _ = ops.head_split(kv[:, C_:], B, heads, 64, Sk, want_x=False, d_src=64)
        scale = 0.125
        Op = torch.empty(M, C_, dtype=BF16, device=dev)
        lse = torch.empty(B, heads, S, dtype=F32, device=dev)
        ops.attn_cross_fwd(Q, K, Vt, Op, lse, B, heads, S, Sk, Skp, 64, scale)
        if T is not None:
            def bwd(dO):
                dq_src = torch.empty(M, C_, dtype=BF16, device=dev)
                dkv = torch.empty_like(kv)
                dQ, dK = torch.empty_like(Q), torch.empty_like(K)
                ops.attn_cross_bwd(Q, K, None, None, kv[:, C_:], Op, dO, lse, dQ, dK, dkv[:, C_:], B, heads, S, Sp, Sk, Skp, 64, scale)
                ops.head_merge(dQ, dq_src, B, heads, 64, S, d_src=64)
                ops.head_merge(dK, dkv[:, :C_], B, heads, 64, Sk, d_src=64)
                return self._linear_bwd(lq, n2, Tl, dq_src), dkv
            T.rec([Op], [n2, kv], bwd)
        return Op

    def _attention_unfused(self, T, qsrc, kvsrc, qo, ko, vo, C_, B, S, Sk, heads):
        """heads wider than 128 (SD1.5: 160 at the 16^2 / 8^2 levels): per (image, head) scores = q k^T (GEMM), row softmax, p v (GEMM); backward
        the same way (two TN GEMMs, two NT GEMMs, the softmax-backward kernel).  The score matrices are tiny at these levels."""
        hd = C_ // heads
        scale = 1.0 / math.sqrt(hd)
        dev = qsrc.device
        self_attn = qsrc is kvsrc
        dp = (hd + 63) // 64 * 64
        Sk8, Sk64, S64 = (Sk + 7) // 8 * 8, (Sk + 63) // 64 * 64, (S + 63) // 64 * 64

        def heads_major(t2d, n, width):               # [B*n, heads*hd] column block -> [B, heads, n_pad, width] zero-padded
            o = torch.zeros(B, heads, (n + 63) // 64 * 64, width, dtype=BF16, device=dev)
            o[:, :, :n, :hd] = t2d.reshape(B, n, heads, hd).permute(0, 2, 1, 3)
            return o

        q3 = heads_major(qsrc[:, qo:qo + C_], S, dp)
        k3 = heads_major(kvsrc[:, ko:ko + C_], Sk, dp)
        v3 = heads_major(kvsrc[:, vo:vo + C_], Sk, dp)
        vt3 = torch.zeros(B, heads, hd, Sk64, dtype=BF16, device=dev)
        vt3[:, :, :, :Sk] = v3[:, :, :Sk, :hd].transpose(2, 3)
        P = torch.zeros(B, heads, S64, Sk64, dtype=BF16, device=dev)
        O = torch.empty(B * S, C_, dtype=BF16, device=dev)
        for b in range(B):
            for h in range(heads):
                sc = P[b, h, :S, :Sk8]
                ops.gemm(q3[b, h, :S], k3[b, h, :Sk8], out=sc)
                if Sk8 != Sk:
                    sc[:, Sk:] = -30000.0
                ops.softmax_rows_(sc, scale)
                ops.gemm(P[b, h, :S], vt3[b, h], out=O[b * S:(b + 1) * S, h * hd:(h + 1) * hd])
        if T is not None:
            def bwd(dO):
                dq_src = torch.zeros_like(qsrc) if self_attn else torch.empty_like(qsrc)
                dkv_src = dq_src if self_attn else torch.zeros_like(kvsrc)
                dO3 = heads_major(dO, S, dp)
                kt3 = torch.zeros(B, heads, hd, Sk64, dtype=BF16, device=dev)
                kt3[:, :, :, :Sk] = k3[:, :, :Sk, :hd].transpose(2, 3)
                dP = torch.zeros(S64, Sk64, dtype=BF16, device=dev)
                for b in range(B):
                    for h in range(heads):
                        p = P[b, h]
                        dv = ops.gemm_tn(p[:, :Sk8], dO3[b, h, :, :hd])                           # [Sk8, hd] = P^T dO
                        ds = dP[:S, :Sk8]
                        ops.gemm(dO3[b, h, :S], v3[b, h, :Sk8], out=ds)                            # dP = dO V^T
                        ops.softmax_rows_bwd_(p[:S, :Sk8], ds, scale)
                        ops.gemm(dP[:S], kt3[b, h], out=dq_src[b * S:(b + 1) * S, qo + h * hd:qo + (h + 1) * hd])      # dQ = dS K
                        dk = ops.gemm_tn(dP[:, :Sk8], q3[b, h, :, :hd])                            # [Sk8, hd] = dS^T Q
                        dkv_src[b * Sk:(b + 1) * Sk, ko + h * hd:ko + (h + 1) * hd] = dk[:Sk]
                        dkv_src[b * Sk:(b + 1) * Sk, vo + h * hd:vo + (h + 1) * hd] = dv[:Sk]
                return (dq_src,) if self_attn else (dq_src, dkv_src)
            T.rec([O], [qsrc] if self_attn else [qsrc, kvsrc], bwd)
        return O

    def _transformer_fwd(self, T, tr, x, ctx2d, B, H, W, Sk):
        S = H * W
        n = self._gn(T, tr.norm, x, B, H, W, 1e-6, silu=False, tokens=True)
        h = self._linear(T, tr.proj_in, n)
        for blk in tr.blocks:
            fused = self._heads_fused_ok(tr.C, tr.heads, B, S)
            n1 = self._ln(T, blk.norm1, h)
            if fused:
                o = self._self_attn_proj_fused(T, blk.qkv, n1, B, S, tr.heads)
            else:
                qkv = self._linear(T, blk.qkv, n1)
                o = self._self_attn(T, qkv, B, S, tr.heads)
            h = self._linear(T, blk.out1, o, residual=h)
            n2 = self._ln(T, blk.norm2, h)
            kv = self._linear(T, blk.kv2, ctx2d, need_dx=False)
            if fused:
                o2 = self._cross_attn_proj_fused(T, blk.q2, n2, kv, B, S, Sk, tr.heads)
            else:
                q = self._linear(T, blk.q2, n2)
                o2 = self._cross_attn(T, q, kv, B, S, Sk, tr.heads)
            h = self._linear(T, blk.out2, o2, residual=h)
            n3 = self._ln(T, blk.norm3, h)
            if blk.ff1.w_il is not None:          # (at every row count: a replica's step and the concatenated batch's must take the same arithmetic path)
                h = self._ffn_geglu_fused(T, blk, n3, h)
                continue
            f = self._linear(T, blk.ff1, n3)
            g = ops.geglu_fwd(f)
            if T is not None:
                T.rec([g], [f], (lambda f_: (lambda dg: (ops.geglu_bwd(f_, dg),)))(f))
            h = self._linear(T, blk.ff2, g, residual=h)
        h = self._linear(T, tr.proj_out, h)
        y = ops.tokens_to_grid(h, B, H, W, residual=x)
        if T is not None:
            T.rec([y], [h, x], lambda dy: (ops.grid_to_tokens(dy, B, H, W), dy))
        return y

    def _ffn_geglu_fused(self, T, blk, n3, h):
        """FeedForward(geglu) of a BasicTransformerBlock with the activation INSIDE its two GEMMs (ST355_EPI_GEGLU / ST355_EPI_GEGLU_GRAD): no [M, 2F] read-back for
        value * gelu(gate), no separate pass for its backward.  Frozen feed-forward weights only (LoRA runs): the projection rows are contracted in the interleaved
        order built once by _prepare_geglu; the weight-gradient path of a full fine-tune keeps the unfused form (its gradients must land in checkpoint order)."""
        l1, l2 = blk.ff1, blk.ff2
        M = n3.shape[0]
        f_il = torch.empty(M, l1.N, dtype=BF16, device=n3.device)
        g = ops.gemm(n3, l1.w_il, bias=l1.b_il, epilogue=EPI_GEGLU, aux_out=f_il)
        y = ops.gemm(g, l2.w.t, bias=None if l2.b is None else l2.b.t, epilogue=EPI_ADD, aux_in=h)
        if T is not None:
            def bwd(dy):
                df = ops.gemm(dy, l2.wT, epilogue=EPI_GEGLU_GRAD, aux_in=f_il)           # [M, 2F] interleaved: d value | d gate
                return ops.gemm(df, l1.wT_il), dy
            T.rec([y], [n3, h], bwd)
        return y

    def _prepare_geglu(self):
        """interleaved copies of every frozen feed-forward projection (ops.geglu_interleave) + their K-major transposes for the dgrad; ST355_GEGLU_FUSED=0 keeps
        the separate st355_geglu_fwd / _bwd passes (A/B switch)"""
        import os
        on = os.environ.get("ST355_GEGLU_FUSED", "1") != "0" and not self.full
        for l in self._all_layers():
            if l.kind == "lin" and getattr(l, "is_geglu_proj", False):
                if on and l.lora is None and l.N % 64 == 0:
                    l.w_il, l.b_il = ops.geglu_interleave(l.w.t, None if l.b is None else l.b.t)
                    l.wT_il = ops.transpose(l.w_il)
                else:
                    l.w_il = l.b_il = l.wT_il = None

    # ------------------------------------------------------------------------------------------------
    def _ckpt_unit(self, T, fn, x, cond):
        """One checkpoint unit = one ResnetBlock2D or one Transformer2DModel (diffusers wraps exactly these when `enable_gradient_checkpointing()` is on:
        the reference's UNet families use diffusers' own flag, common.py:3560-3636; interval / stride select units as `should_checkpoint_block` does).
        Checkpointed: the unit runs WITHOUT a tape, and ONE record stands for it whose backward re-runs it on a private tape — same kernels, same order,
        same values: bit-identical gradients — and sweeps that tape.  fn(T, x, cond) -> y; gradients flow to x and cond."""
        idx = self._unit_counter
        self._unit_counter += 1
        k, s_ = self.gradient_checkpointing_interval, self.gradient_checkpointing_segment_stride
        on = self.gradient_checkpointing and (k is None or k <= 1 or (idx % k == 0 if s_ is None else idx % s_ < k))
        if T is None:
            return fn(None, x, cond)
        if not on:
            # kept unit: recorded on a private tape as well, so that its input gradients are summed in the SAME association as in the recomputed form
            # (bf16 sums are not associative: x also feeds skip connections) — checkpointed and direct runs then agree bit for bit
            Tk = Tape()
            yk = fn(Tk, x, cond)
            T.rec([yk], [x, cond], lambda dy: tuple(Tk.backward(yk, dy, wrt=[x, cond])))
            return yk
        y = fn(None, x, cond)

        def bwd(dy):
            T2 = Tape()
            y2 = fn(T2, x, cond)
            return tuple(T2.backward(y2, dy, wrt=[x, cond]))
        T.rec([y], [x, cond], bwd)
        return y

    def _engine_forward(self, sample, timestep, ehs, text_embeds, time_ids, save: bool):
        c = self.config
        dev = self.device_
        T = Tape() if save else None
        self._unit_counter = 0
        for g_ in self.lora_groups:
            g_.pack()
        B, Cin, H, W = sample.shape
        nb = len(c.block_out_channels)
        # ---- embeddings ----
        t32 = timestep.to(device=dev, dtype=F32).reshape(-1).expand(B).contiguous()
        t_emb = ops.timestep_proj(t32, c.block_out_channels[0], 1.0)
        e1 = self._linear(T, self.t1, t_emb, need_dx=False)
        s1 = ops.silu(e1)
        if T is not None:
            T.rec([s1], [e1], lambda d: (ops.silu_bwd(e1, d),))
        emb = self._linear(T, self.t2, s1)
        if c.addition_embed_type == "text_time":
            tid = ops.timestep_proj(time_ids.to(device=dev, dtype=F32).reshape(-1).contiguous(), c.addition_time_embed_dim, 1.0).view(B, -1)
            add = torch.cat([text_embeds.to(device=dev, dtype=BF16), tid], dim=1).contiguous()
            a1 = self._linear(T, self.a1, add, need_dx=False)
            sa = ops.silu(a1)
            if T is not None:
                T.rec([sa], [a1], lambda d: (ops.silu_bwd(a1, d),))
            a2 = self._linear(T, self.a2, sa)
            emb2 = ops.add(emb, a2)
            if T is not None:
                T.rec([emb2], [emb, a2], lambda d: (d, d))
            emb = emb2
        se = ops.silu(emb)
        if T is not None:
            T.rec([se], [emb], (lambda e_: (lambda d: (ops.silu_bwd(e_, d),)))(emb))
        Sk = ehs.shape[1]
        ctx2d = ehs.to(device=dev, dtype=BF16).reshape(B * Sk, -1).contiguous()
        # ---- conv_in (column path: 4 -> 8 channels, K = 72 -> 128) ----
        col = ops.im2col3x3(ops.grid_from_nchw(sample.to(BF16), 8), B, H, W, stride=1)
        x = ops.conv(col, self.l_conv_in.w.t, B, H, W, bias=self.l_conv_in.b.t, taps=1)
        if T is not None:
            def bwd_in(dy, col=col, H=H, W=W):
                if self.full:
                    ops.conv_wgrad(col, dy, self.l_conv_in.w.g, B, H, W, taps=1)
                    self._bias_grad(dy, self.l_conv_in.b, rows=B * (H + 2) * (W + 2))
                    self._ready(self.l_conv_in.b, self.l_conv_in.w)
                return (None,)
            T.rec([x], [None], bwd_in)
        skips = [x]
        h_, w_ = H, W
        for i, blk in enumerate(self.down):
            for j, r in enumerate(blk.resnets):
                x = self._ckpt_unit(T, lambda T_, x_, se_, r=r, h_=h_, w_=w_: self._resnet_fwd(T_, r, x_, se_, B, h_, w_), x, se)
                if blk.attns:
                    x = self._ckpt_unit(T, lambda T_, x_, c_, a=blk.attns[j], h_=h_, w_=w_: self._transformer_fwd(T_, a, x_, c_, B, h_, w_, Sk), x, ctx2d)
                skips.append(x)
            if blk.down is not None:
                x = self._downsample(T, blk.down, x, B, h_, w_)
                h_, w_ = h_ // 2, w_ // 2
                skips.append(x)
        x = self._ckpt_unit(T, lambda T_, x_, se_, h_=h_, w_=w_: self._resnet_fwd(T_, self.mid.r0, x_, se_, B, h_, w_), x, se)
        x = self._ckpt_unit(T, lambda T_, x_, c_, h_=h_, w_=w_: self._transformer_fwd(T_, self.mid.attn, x_, c_, B, h_, w_, Sk), x, ctx2d)
        x = self._ckpt_unit(T, lambda T_, x_, se_, h_=h_, w_=w_: self._resnet_fwd(T_, self.mid.r1, x_, se_, B, h_, w_), x, se)
        for i, blk in enumerate(self.up):
            for j, r in enumerate(blk.resnets):
                sk = skips.pop()
                c1 = x.shape[1]
                cat = torch.cat([x, sk], dim=1)
                if T is not None:
                    T.rec([cat], [x, sk], (lambda c1_: (lambda d: (d[:, :c1_].contiguous(), d[:, c1_:].contiguous())))(c1))
                x = self._ckpt_unit(T, lambda T_, x_, se_, r=r, h_=h_, w_=w_: self._resnet_fwd(T_, r, x_, se_, B, h_, w_), cat, se)
                if blk.attns:
                    x = self._ckpt_unit(T, lambda T_, x_, c_, a=blk.attns[j], h_=h_, w_=w_: self._transformer_fwd(T_, a, x_, c_, B, h_, w_, Sk), x, ctx2d)
            if blk.up is not None:
                u = ops.upsample2x(x, B, h_, w_)
                if T is not None:
                    T.rec([u], [x], (lambda hh, ww: (lambda d: (ops.upsample2x_bwd(d, B, hh, ww),)))(h_, w_))
                h_, w_ = 2 * h_, 2 * w_
                x = self._conv3(T, blk.up, u, B, h_, w_)
        x = self._gn(T, self.norm_out, x, B, h_, w_, c.norm_eps)
        yg = ops.conv(x, self.l_conv_out.w.t, B, h_, w_, bias=self.l_conv_out.b.t)
        out = ops.grid_to_nchw(yg, B, c.out_channels, h_, w_)
        if T is not None:
            def bwd_out(dout, x=x, H=h_, W=w_):
                l = self.l_conv_out
                dyg = ops.grid_from_nchw(dout.to(BF16).contiguous(), 8)
                if self.full:
                    ops.conv_wgrad(x, dyg, l.w.g, B, H, W, taps=9)
                    self._bias_grad(dyg, l.b, rows=B * (H + 2) * (W + 2))
                    self._ready(l.b, l.w)
                dcol = ops.im2col3x3(dyg, B, H, W, stride=1)
                return (ops.conv(dcol, l.wT, B, H, W, taps=1),)
            T.rec([out], [x], bwd_out)
        return out, T

    def _downsample(self, T, l, x, B, H, W):
        col = ops.im2col3x3(x, B, H, W, stride=2)
        y = ops.conv(col, l.w.t, B, H // 2, W // 2, bias=l.b.t, taps=1)
        if T is not None:
            n = B * (H // 2 + 2) * (W // 2 + 2)

            def bwd(dy):
                if self.full:
                    ops.conv_wgrad(col, dy, l.w.g, B, H // 2, W // 2, taps=1)
                    self._bias_grad(dy, l.b, rows=n)
                    self._ready(l.b, l.w)
                dcol = torch.empty(dy.shape[0], col.shape[1], dtype=BF16, device=dy.device)
                ops.gemm(dy[:n], l.wT, out=dcol[:n])
                return (ops.col2im3x3(dcol, B, H, W, x.shape[1], stride=2),)
            T.rec([y], [x], bwd)
        return y

    # ------------------------------------------------------------------------------------------------
    # public forward (diffusers UNet2DConditionModel.forward as the reference calls it: sdxl/model.py:350-367)
    # ------------------------------------------------------------------------------------------------
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, added_cond_kwargs=None, return_dict: bool = True,
                cross_attention_kwargs=None, **unsupported):
        for k, v in unsupported.items():
            if v is not None and v is not False:
                raise NotImplementedError(f"UNet2DConditionModel(st355): argument {k!r} is not supported on the HIP path")
        te = ti = None
        if self.config.addition_embed_type == "text_time":
            if not added_cond_kwargs or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
                raise ValueError("addition_embed_type 'text_time' needs added_cond_kwargs['text_embeds'] and ['time_ids']")
            te, ti = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
        if torch.is_grad_enabled() and (self.full or self._lora_params):
            dummy = te if te is not None else encoder_hidden_states
            out = _UNetFn.apply(self, sample, timestep, encoder_hidden_states, dummy, ti, *(self._full_params if self.full else self._lora_params))
        else:
            with torch.no_grad():
                out, _ = self._engine_forward(sample, timestep, encoder_hidden_states, te, ti, save=False)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)


class _UNetFn(torch.autograd.Function):
    """one autograd node for the whole network; backward fills the bf16 gradient arena and hands autograd views of a private copy"""

    @staticmethod
    def forward(fctx, model, sample, timestep, ehs, te, ti, *params):
        has_add = model.config.addition_embed_type == "text_time"
        out, tape = model._engine_forward(sample.detach(), timestep.detach(), ehs.detach(), te.detach() if has_add else None,
                                          ti.detach() if has_add else None, save=True)
        fctx.model, fctx.tape, fctx.out = model, tape, out
        return out

    @staticmethod
    def backward(fctx, dout):
        model = fctx.model
        if model.grad_sync is not None:
            model.grad_sync.begin()
        if model.full:
            model._refresh_transposed()
        fctx.tape.backward(fctx.out, dout.contiguous())
        fctx.tape = None
        arena = model.grad_arena if model.full else model.lora_grad_flat
        if model.grad_sync is not None:
            if not model.full:
                model.grad_sync.ready(0, arena.numel())          # LoRA: one small all-reduce after the backward
            model.grad_scale_from_sync = model.grad_sync.finish()
            sent = sum(hi - lo for lo, hi in model.grad_sync.launched_slices)
            if sent != arena.numel():
                raise RuntimeError(f"gradient sync covered {sent} of {arena.numel()} elements: a parameter's gradient was never reported ready")
        from ..training.grad_sync import hand_over_gradients
        gflat = hand_over_gradients(model, arena)
        model._last_grad_flat = gflat
        if model.full:
            grads = [gflat[off:off + n].view_as(p) for (off, n), p in zip(model._full_offsets, model._full_params)]
        else:
            grads, off = [], 0
            for p in model._lora_params:
                grads.append(gflat[off:off + p.numel()].view_as(p))
                off += p.numel()
        return (None,) * 6 + tuple(grads)
