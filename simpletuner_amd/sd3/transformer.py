"""SD3Transformer2DModel — the MI355X-native trained component for the SD3 MMDiT (joint transformer blocks).

Mirrors the reference's module surface (simpletuner/helpers/models/sd3/transformer.py:244-911): same constructor arguments,
`forward(hidden_states[B,16,H,W], encoder_hidden_states, pooled_projections, timestep (0..1000), return_dict)` contract, the
diffusers state-dict keys (`pos_embed.proj.weight` stays the Conv2d [D,C,p,p] tensor; `pos_embed.pos_embed` the sincos buffer),
`.config`, `.parameters()`.  Forward and the hand-written backward are sequences of libst355 launches — the same kernels as
the Flux double block at D = 1536 / head_dim 64:
  * PatchEmbed = 2x2 patchify (im2col order) + ONE GEMM whose epilogue adds bias and the centre-cropped position table;
  * joint attention over [sample || context] (sd3 order, `JointAttnProcessor2_0`), no RoPE (identity tables), optional q/k RMSNorm;
  * the last block is `context_pre_only`: its context stream only feeds K/V/Q (AdaLayerNormContinuous, chunk order scale,shift);
  * every AdaLN modulation linear of all blocks (+norm_out) is one matrix -> one skinny GEMM per step;
  * unpatchify "nhwpqc->nchpwq" is the order-1 permute kernel.
Two training modes: frozen-base LoRA (`add_lora_adapter`) and the full fine-tune of BASELINE.json configs[3] (`enable_full_finetune`: TN weight-gradient
GEMMs, token-axis reductions for the modulation rows, q / k RMSNorm weight gradients; `_engine_backward_full`).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..flux.transformer import FluxTransformer2DModel as _FluxEngine, LoraGroup, _attach, _frozen
from ..ops import EPI_ADD, EPI_GATE_RESIDUAL, EPI_GELU, EPI_MUL_GELU_GRAD
from ..training.checkpoint_plan import CheckpointPlanMixin

BF16 = torch.bfloat16
F32 = torch.float32
_BLOCK_ABI = __import__("os").environ.get("ST355_BLOCK_ABI", "1") != "0"      # A/B switch: 0 = sequence the blocks' kernels from the host instead of st355_block_sd3_joint_*


def sincos_2d(embed_dim: int, grid_size: int, base_size: int, interpolation_scale: float = 1.0) -> torch.Tensor:
    """diffusers get_2d_sincos_pos_embed: [grid_size^2, embed_dim] fp32 (w axis first, sin then cos per axis)"""
    ar = torch.arange(grid_size, dtype=torch.float32) / (grid_size / base_size) / interpolation_scale
    gw = ar[None, :].expand(grid_size, grid_size).reshape(-1).double()      # meshgrid(grid_w, grid_h): w varies fastest
    gh = ar[:, None].expand(grid_size, grid_size).reshape(-1).double()

    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
        out = pos[:, None] * omega[None, :]
        return torch.cat([out.sin(), out.cos()], dim=1)

    return torch.cat([one_d(embed_dim // 2, gw), one_d(embed_dim // 2, gh)], dim=1).float()



def _pad64_empty(rows: int, cols: int, dev):
    """[rows, cols] bf16, uninitialised, as the head of a parent buffer whose row count is rounded up to 64 and whose tail rows are ZERO: the weight-gradient GEMM
    (contraction granule 64 rows) reads the parent directly instead of a zero-padded copy of the tensor (round 5: two launches and a full copy per text-stream
    operand, 192 hipMemcpy + 224 fill launches per SD3 full fine-tune step).  The view remembers its parent in `_st355_pad64`."""
    rp = (rows + 63) // 64 * 64
    if rp == rows:
        return torch.empty(rows, cols, dtype=BF16, device=dev)
    par = torch.empty(rp, cols, dtype=BF16, device=dev)
    par[rows:].zero_()
    t = par[:rows]
    t._st355_pad64 = par
    return t


def _rows3(joint, lo: int, rows: int, B: int, S: int):
    """rows [lo, lo + rows) of every sample of a joint [B * S, C] buffer as a GEMM operand ([B, rows, C] strided view, no copy)"""
    return _FluxEngine._rows_of(joint, lo, rows, SimpleNamespace(B=B, S=S))


def _stream_problems(B: int, S: int, rows: int, pr: dict, after: Optional[list] = None):
    """one projection over the `rows`-row block of every sample as ONE problem: segmented operands when the block is tile-aligned (the 4096 image rows,
    FluxTransformer2DModel._problems); otherwise (the 154 text rows) through compact copies of the joint-buffer operands — the rows are gathered before the
    GEMM, a joint-buffer `out` is written to a compact temporary and scattered back by the closures appended to `after` (run them once the launch is
    issued).  A few MB of copies instead of B tiny launches per projection (measured: the per-sample form of the text stream cost as much as the image
    stream's segmentation saved)."""
    if B == 1 or rows % 256 == 0 or after is None or rows >= 1024:       # big unaligned blocks (odd aspect buckets): one problem per sample beats copying them
        return _FluxEngine._problems(SimpleNamespace(B=B, S=S), rows, pr)
    q = dict(pr)
    for k in ("a", "a2", "aux_in"):
        v = q.get(k)
        if torch.is_tensor(v) and v.dim() == 3:
            q[k] = v.reshape(B * rows, v.shape[-1])                         # gather (copy)
    for k in ("out", "aux_out"):
        v = q.get(k)
        if torch.is_tensor(v) and v.dim() == 3:
            tmp = torch.empty(B * rows, v.shape[-1], dtype=v.dtype, device=v.device)
            q[k] = tmp
            after.append(lambda dst=v, src=tmp: dst.copy_(src.view(B, rows, -1)))      # scatter back into the joint rows
    return [q]



class SD3Transformer2DModel(CheckpointPlanMixin, nn.Module):
    def __init__(self, sample_size: int = 128, patch_size: int = 2, in_channels: int = 16, num_layers: int = 18,
                 attention_head_dim: int = 64, num_attention_heads: int = 18, joint_attention_dim: int = 4096,
                 caption_projection_dim: int = 1152, pooled_projection_dim: int = 2048, out_channels: int = 16,
                 pos_embed_max_size: int = 96, dual_attention_layers: Tuple[int, ...] = (), qk_norm: Optional[str] = None,
                 device=None, **_ignored):
        super().__init__()
        if patch_size != 2:
            raise ValueError("patch_size must be 2 (the patchify kernels are 2x2)")
        self.dual_layers = frozenset(int(i) for i in (dual_attention_layers or ()))
        if any(i < 0 or i >= num_layers - 1 for i in self.dual_layers):
            raise ValueError("dual_attention_layers must name regular (not context_pre_only) blocks")      # sd3/transformer.py:295: the last block never is
        if attention_head_dim not in (64, 128):
            raise ValueError("attention_head_dim must be 64 or 128 (kernels built for these)")
        if qk_norm not in (None, "rms_norm"):
            raise ValueError("qk_norm must be None or 'rms_norm'")
        self.H, self.hd = num_attention_heads, attention_head_dim
        self.D = D = self.H * self.hd
        self.inner_dim = D
        if caption_projection_dim != D:
            raise ValueError("caption_projection_dim must equal num_attention_heads * attention_head_dim")
        self.config = SimpleNamespace(sample_size=sample_size, patch_size=patch_size, in_channels=in_channels, num_layers=num_layers,
                                      attention_head_dim=attention_head_dim, num_attention_heads=num_attention_heads,
                                      joint_attention_dim=joint_attention_dim, caption_projection_dim=caption_projection_dim,
                                      pooled_projection_dim=pooled_projection_dim, out_channels=out_channels,
                                      pos_embed_max_size=pos_embed_max_size, dual_attention_layers=tuple(sorted(self.dual_layers)), qk_norm=qk_norm)
        self.out_channels = out_channels
        if D % 64 or joint_attention_dim % 64 or pooled_projection_dim % 64 or (4 * in_channels) % 64:
            raise ValueError("all contraction dims must be multiples of 64")
        self.device_ = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        dev = self.device_
        # every weight / bias / modulation row is a view of ONE bf16 arena (allocation order = arena order): the full fine-tune then
        # has one gradient arena of the same layout, ONE fused optimizer launch and contiguous slices for the RCCL all-reduce.
        # Pass 1 counts (meta tensors), pass 2 hands out the views.
        self._arena_numel = 0
        self._counting = True
        self._build(lambda *s_: self._alloc(s_))
        self.arena = torch.zeros(self._arena_numel, dtype=BF16, device=dev)
        self._arena_numel = 0
        self._counting = False
        self._build(lambda *s_: self._alloc(s_))
        self.pos_embed.register_buffer("pos_embed", torch.zeros(1, pos_embed_max_size * pos_embed_max_size, D, dtype=F32, device=dev))

        self.lora_groups: List[LoraGroup] = []
        self.lora_flat: Optional[torch.Tensor] = None
        self.lora_grad_flat: Optional[torch.Tensor] = None
        self._lora_params: List[nn.Parameter] = []
        self._cache: Dict = {}
        self._prepared = False
        self.accumulate_lora_grads = False
        self.gradient_checkpointing = False
        self.gradient_checkpointing_interval = None
        self.gradient_checkpointing_segment_stride = None
        self._tread_router, self._tread_routes = None, None
        self.grad_sync = None
        self._last_grad_flat = None
        self.full = False

    def _alloc(self, shape):
        n = 1
        for d in shape:
            n *= d
        n_pad = (n + 7) // 8 * 8                      # every tensor starts 16-byte aligned inside the arena
        off = self._arena_numel
        self._arena_numel += n_pad
        if self._counting:
            return torch.empty(*shape, dtype=BF16, device="meta")
        return self.arena[off:off + n].view(*shape)

    def _reg(self, name, param):
        if not self._counting:
            _attach(self, name, param)

    def _build(self, e):
        c = self.config
        D, dev = self.D, self.device_
        in_channels, out_channels, num_layers, qk_norm = c.in_channels, c.out_channels, c.num_layers, c.qk_norm
        joint_attention_dim, pooled_projection_dim = c.joint_attention_dim, c.pooled_projection_dim

        def lin(name, out_f, in_f):
            w, b = e(out_f, in_f), e(out_f)
            self._reg(name + ".weight", _frozen(w)); self._reg(name + ".bias", _frozen(b))
            return SimpleNamespace(w=w, b=b, wT=None, lora=None)

        # PatchEmbed: the Conv2d weight keeps its checkpoint shape; the GEMM reads it as [D, C*p*p]
        conv_w, conv_b = e(D, in_channels, 2, 2), e(D)
        self._reg("pos_embed.proj.weight", _frozen(conv_w)); self._reg("pos_embed.proj.bias", _frozen(conv_b))
        self.l_patch = SimpleNamespace(w=conv_w.view(D, 4 * in_channels), b=conv_b)
        self.l_t1 = lin("time_text_embed.timestep_embedder.linear_1", D, 256)
        self.l_t2 = lin("time_text_embed.timestep_embedder.linear_2", D, D)
        self.l_p1 = lin("time_text_embed.text_embedder.linear_1", D, pooled_projection_dim)
        self.l_p2 = lin("time_text_embed.text_embedder.linear_2", D, D)
        self.l_ctx = lin("context_embedder", D, joint_attention_dim)

        # last block: norm1_context is AdaLayerNormContinuous (2D); a dual-attention block's norm1 is SD35AdaLayerNormZeroX (9 chunks instead of 6)
        self.mod_total = (num_layers * 12 - 4 + 2 + 3 * len(self.dual_layers)) * D
        self.mod_w, self.mod_b = e(self.mod_total, D), e(self.mod_total)
        off = 0

        def mod_slice(name, n):
            nonlocal off
            self._reg(name + ".weight", _frozen(self.mod_w[off:off + n])); self._reg(name + ".bias", _frozen(self.mod_b[off:off + n]))
            o = off
            off += n
            return o

        def fused(prefix, names, out_each, in_f):
            n = len(names)
            w, b = e(n * out_each, in_f), e(n * out_each)
            for j, nm in enumerate(names):
                self._reg(f"{prefix}{nm}.weight", _frozen(w[j * out_each:(j + 1) * out_each]))
                self._reg(f"{prefix}{nm}.bias", _frozen(b[j * out_each:(j + 1) * out_each]))
            return SimpleNamespace(w=w, b=b, wT=None, lora=None)

        def normw(name):
            if qk_norm is None:
                return None
            w = e(self.hd)
            if not self._counting:
                w.fill_(1.0)
            self._reg(name + ".weight", _frozen(w))
            return w

        self.blocks: List[SimpleNamespace] = []
        self._blocks_arena_lo = self._arena_numel
        for i in range(num_layers):
            p = f"transformer_blocks.{i}."
            blk = SimpleNamespace(last=(i == num_layers - 1), arena_lo=self._arena_numel, dual=(i in self.dual_layers))
            blk.mod_off = mod_slice(p + "norm1.linear", (9 if blk.dual else 6) * D)
            blk.mod_off_c = mod_slice(p + "norm1_context.linear", (2 if blk.last else 6) * D)
            blk.qkv = fused(p + "attn.", ["to_q", "to_k", "to_v"], D, D)
            blk.add_qkv = fused(p + "attn.", ["add_q_proj", "add_k_proj", "add_v_proj"], D, D)
            blk.to_out = fused(p + "attn.", ["to_out.0"], D, D)
            blk.to_add_out = None if blk.last else fused(p + "attn.", ["to_add_out"], D, D)
            blk.norm_q, blk.norm_k = normw(p + "attn.norm_q"), normw(p + "attn.norm_k")
            blk.norm_added_q, blk.norm_added_k = normw(p + "attn.norm_added_q"), normw(p + "attn.norm_added_k")
            # SD3.5 dual attention (sd3/transformer.py:155-165, 190-197): a second, image-only self-attention fed by the 7th-9th modulation chunks
            blk.qkv2 = fused(p + "attn2.", ["to_q", "to_k", "to_v"], D, D) if blk.dual else None
            blk.to_out2 = fused(p + "attn2.", ["to_out.0"], D, D) if blk.dual else None
            blk.norm_q2, blk.norm_k2 = (normw(p + "attn2.norm_q"), normw(p + "attn2.norm_k")) if blk.dual else (None, None)
            blk.ff1 = fused(p, ["ff.net.0.proj"], 4 * D, D)
            blk.ff2 = fused(p, ["ff.net.2"], D, 4 * D)
            blk.ffc1 = None if blk.last else fused(p, ["ff_context.net.0.proj"], 4 * D, D)
            blk.ffc2 = None if blk.last else fused(p, ["ff_context.net.2"], D, 4 * D)
            blk.arena_hi = self._arena_numel
            self.blocks.append(blk)
        self.mod_off_out = mod_slice("norm_out.linear", 2 * D)
        assert off == self.mod_total
        self._head_arena_lo = self._arena_numel
        self.l_out = lin("proj_out", 4 * out_channels, D)


    # ------------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def load_flat_state(self, state: Dict[str, torch.Tensor]):
        own = dict(self.named_parameters())
        missing = [k for k in own if k not in state and ".lora_" not in k]
        if missing:
            raise KeyError(f"missing weights: {missing[:5]} ... ({len(missing)})")
        for k, v in state.items():
            if k in own:
                own[k].data.copy_(v.to(device=own[k].device, dtype=own[k].dtype))
        if "pos_embed.pos_embed" in state:
            self.pos_embed.pos_embed.copy_(state["pos_embed.pos_embed"].to(self.device_, F32))
        self._prepared = False
        self._cache.clear()

    @torch.no_grad()
    def init_synthetic(self, seed: int = 42):
        g = torch.Generator(device=self.device_).manual_seed(seed)
        for name, p in self.named_parameters():
            if ".lora_" in name:
                continue
            if "norm_q" in name or "norm_k" in name or "norm_added" in name:
                p.data.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=self.device_))
            elif name.endswith(".bias"):
                p.data.copy_(0.02 * torch.randn(p.shape, generator=g, device=self.device_))
            else:
                fan_in = p[0].numel()
                p.data.copy_(torch.randn(p.shape, generator=g, device=self.device_, dtype=BF16) * (1.0 / math.sqrt(fan_in)))
        c = self.config
        self.pos_embed.pos_embed.copy_(sincos_2d(self.D, c.pos_embed_max_size, c.sample_size // c.patch_size)[None].to(self.device_))
        self._prepared = False
        self._cache.clear()

    @torch.no_grad()
    def prepare_for_training(self):
        """K-major transposed copies for the dgrad GEMMs (frozen base => built once)"""
        for blk in self.blocks:
            for l in (blk.qkv, blk.add_qkv, blk.to_out, blk.to_add_out, blk.qkv2, blk.to_out2, blk.ff1, blk.ff2, blk.ffc1, blk.ffc2):
                if l is not None:
                    l.wT = l.w.t().contiguous()
        self.l_out.wT = self.l_out.w.t().contiguous()
        self._prepared = True

    # ------------------------------------------------------------------------------------------------
    # LoRA (peft naming), sd3/model.py:122 DEFAULT_LORA_TARGET = to_k,to_q,to_v,to_out.0
    # ------------------------------------------------------------------------------------------------
    def add_lora_adapter(self, rank: int = 32, alpha: Optional[float] = None, targets: str = "default", seed: int = 7, init_b_std: float = 0.0):
        alpha = float(rank if alpha is None else alpha)
        D, dev = self.D, self.device_
        plan = []
        self.lora_groups = []

        def group(lin, prefix, names, K):
            g = LoraGroup(K, lin.w.shape[0], [(prefix + n, j * (lin.w.shape[0] // len(names)), lin.w.shape[0] // len(names))
                                              for j, n in enumerate(names)], rank, alpha, dev)
            lin.lora = g
            self.lora_groups.append(g)
            for (name, n_off, N) in g.targets:
                plan.append((g, name, N, K))

        for i, blk in enumerate(self.blocks):
            p = f"transformer_blocks.{i}.attn."
            group(blk.qkv, p, ["to_q", "to_k", "to_v"], D)
            group(blk.to_out, p, ["to_out.0"], D)
            if blk.dual:       # peft matches target modules by suffix: attn2.to_q / to_k / to_v / to_out.0 carry adapters too
                p2 = f"transformer_blocks.{i}.attn2."
                group(blk.qkv2, p2, ["to_q", "to_k", "to_v"], D)
                group(blk.to_out2, p2, ["to_out.0"], D)
            if targets == "all":
                group(blk.add_qkv, p, ["add_q_proj", "add_k_proj", "add_v_proj"], D)
                if blk.to_add_out is not None:
                    group(blk.to_add_out, p, ["to_add_out"], D)
        total = sum(rank * K + N * rank for (_, _, N, K) in plan)
        total = (total + 7) // 8 * 8
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
            bound = 1.0 / math.sqrt(K)
            a.copy_((torch.rand(rank, K, generator=gen, device=dev) * 2 - 1) * bound)
            if init_b_std > 0:
                b.copy_(torch.randn(N, rank, generator=gen, device=dev) * init_b_std)
            pa, pb = nn.Parameter(a), nn.Parameter(b)
            _attach(self, name + ".lora_A.default.weight", pa); _attach(self, name + ".lora_B.default.weight", pb)
            g.A.append(pa.data); g.B.append(pb.data); g.gA.append(ga); g.gB.append(gb)
            g.flat_hi = off
            self._lora_params += [pa, pb]
        return self._lora_params

    # ------------------------------------------------------------------------------------------------
    def _tables(self, h: int, w: int, S: int, B: int):
        """cropped position table expanded over the batch ([B*h*w, D] bf16) and identity RoPE tables (SD3 has no RoPE)"""
        key = (h, w, S, B)
        hit = self._cache.get(key)
        if hit is None:
            mx = self.config.pos_embed_max_size
            if h > mx or w > mx:
                raise ValueError(f"latent grid {h}x{w} exceeds pos_embed_max_size {mx}")
            top, left = (mx - h) // 2, (mx - w) // 2
            pos = self.pos_embed.pos_embed[0].view(mx, mx, self.D)[top:top + h, left:left + w].reshape(h * w, self.D).to(BF16)
            pos = pos[None].expand(B, -1, -1).reshape(B * h * w, self.D).contiguous()
            cos = torch.ones(S, self.hd, dtype=F32, device=self.device_); sin = torch.zeros(S, self.hd, dtype=F32, device=self.device_)
            hit = (pos, cos, sin)
            self._cache[key] = hit
        return hit

    def _block_fwd(self, blk, img, txt, env, save: bool):
        """one JointTransformerBlock (sd3/transformer.py:145-241 `_sd3_apply_joint_transformer_block`): returns (img', txt' | None, saved activations | None).
        Also the RECOMPUTE unit of the activation-checkpoint plans (training/checkpoint_plan.py): run again from a segment's kept input in backward, it
        launches the same kernels in the same order on the same values — bit-identical activations, hence bit-identical gradients."""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        B, Si, St, S, Sp, cos, sin, mod, scale, full = env.B, env.Si, env.St, env.S, env.Sp, env.cos, env.sin, env.mod, env.scale, env.full
        ybuf = (lambda rows: torch.empty(rows, D, dtype=BF16, device=dev)) if (full and save) else (lambda rows: None)
        rpb = 1 if env.rpb_i == 1 else Si     # rows of the image stream that share one modulation row: the sample's (inside a TREAD route: its kept) rows, or 1 under tokenwise timesteps
        if _BLOCK_ABI and not blk.dual and ops.ATTN_TR and img.is_contiguous() and txt.is_contiguous() and rpb == Si:
            return self._block_fwd_c(blk, img, txt, env, save, ybuf)

        mi = env.mod_i[:, blk.mod_off:blk.mod_off + 6 * D]
        n_img = ops.ln_modulate_fwd(img, mi[:, D:2 * D], mi[:, :D], rpb)
        if blk.last:
            mt = mod[:, blk.mod_off_c:blk.mod_off_c + 2 * D]
            n_txt = ops.ln_modulate_fwd(txt, mt[:, :D], mt[:, D:2 * D], St)          # AdaLayerNormContinuous: (scale, shift)
        else:
            mt = mod[:, blk.mod_off_c:blk.mod_off_c + 6 * D]
            n_txt = ops.ln_modulate_fwd(txt, mt[:, D:2 * D], mt[:, :D], St)
        qkv = torch.empty(B * S, 3 * D, dtype=BF16, device=dev)
        T_img = ops.gemm(n_img, blk.qkv.lora.A_cat) if blk.qkv.lora is not None else None
        T_txt = ops.gemm(n_txt, blk.add_qkv.lora.A_cat) if blk.add_qkv.lora is not None else None
        kw_i = dict(a2=T_img, b2=blk.qkv.lora.B_blk) if T_img is not None else {}
        kw_t = dict(a2=T_txt, b2=blk.add_qkv.lora.B_blk) if T_txt is not None else {}
        # the image stream (4096 rows per sample: tile-aligned) is ONE segmented problem over the joint buffer; the 154 text rows stay per sample
        after = []
        ops.gemm_grouped(_stream_problems(B, S, Si, dict(a=n_img, w=blk.qkv.w, bias=blk.qkv.b, out=_rows3(qkv, 0, Si, B, S), **kw_i), after)
                         + _stream_problems(B, S, St, dict(a=n_txt, w=blk.add_qkv.w, bias=blk.add_qkv.b, out=_rows3(qkv, Si, St, B, S), **kw_t), after))
        for f in after:
            f()
        Q = torch.empty(B, H, S, hd, dtype=BF16, device=dev); K = torch.empty_like(Q)
        mk = torch.zeros if Sp > S else torch.empty
        # head-major Q^T / K^T copies only for the copy-reading backward kernels (ST355_ATTN_TR=0); the default backward gathers those fragments by
        # transposing LDS reads from the row-major tiles (attention_bwd.hip dkv3 / dq<TR>, r3: head_dim 64 too)
        Qt = Kt = None
        if not ops.ATTN_TR:
            Qt = mk(B, H, hd, Sp, dtype=BF16, device=dev); Kt = mk(B, H, hd, Sp, dtype=BF16, device=dev)
        Vt = mk(B, H, hd, Sp, dtype=BF16, device=dev)
        ops.qk_norm_rope_fwd(qkv, blk.norm_q, blk.norm_k, cos, sin, Q, K, Qt, Kt, Vt, B, H, hd, Si, 0, S, Sp)
        ops.qk_norm_rope_fwd(qkv, blk.norm_added_q, blk.norm_added_k, cos, sin, Q, K, Qt, Kt, Vt, B, H, hd, St, Si, S, Sp)
        O = torch.empty(B * S, D, dtype=BF16, device=dev); lse2 = torch.empty(B, H, S, dtype=F32, device=dev)
        ops.attn_fwd(Q, K, Vt, O, lse2, B, H, S, Sp, hd, scale)
        del Vt
        x1_img = torch.empty(B * Si, D, dtype=BF16, device=dev)
        x1_txt = None if blk.last else torch.empty(B * St, D, dtype=BF16, device=dev)
        T_o = torch.empty(B * Si, blk.to_out.lora.K2, dtype=BF16, device=dev) if blk.to_out.lora is not None else None
        T_ao = (torch.empty(B * St, blk.to_add_out.lora.K2, dtype=BF16, device=dev)
                if (not blk.last and blk.to_add_out.lora is not None) else None)
        ya_i, ya_t = ybuf(B * Si), (None if blk.last else ybuf(B * St))
        O_i, O_t = _rows3(O, 0, Si, B, S), _rows3(O, Si, St, B, S)
        kw_i, kw_t = {}, {}
        if ya_i is not None:
            kw_i["aux_out"] = ya_i
        if ya_t is not None:
            kw_t["aux_out"] = ya_t
        after = []
        if B > 1 and St % 256:
            O_t = O_t.reshape(B * St, D)                                  # the text rows of the attention output, gathered once for both uses below
        if T_o is not None:
            for pr in _stream_problems(B, S, Si, dict(a=O_i, w=blk.to_out.lora.A_cat, out=T_o), after):
                ops.gemm(pr.pop("a"), pr.pop("w"), **pr)
            kw_i.update(a2=T_o, b2=blk.to_out.lora.B_blk)
        probs = _stream_problems(B, S, Si, dict(a=O_i, w=blk.to_out.w, bias=blk.to_out.b, out=x1_img, epilogue=EPI_GATE_RESIDUAL, aux_in=img,
                                                gate=mi[:, 2 * D:3 * D], rows_per_batch=rpb, **kw_i), after)
        if not blk.last:
            if T_ao is not None:
                for pr in _stream_problems(B, S, St, dict(a=O_t, w=blk.to_add_out.lora.A_cat, out=T_ao), after):
                    ops.gemm(pr.pop("a"), pr.pop("w"), **pr)
                kw_t.update(a2=T_ao, b2=blk.to_add_out.lora.B_blk)
            probs += _stream_problems(B, S, St, dict(a=O_t, w=blk.to_add_out.w, bias=blk.to_add_out.b, out=x1_txt, epilogue=EPI_GATE_RESIDUAL,
                                                     aux_in=txt, gate=mt[:, 2 * D:3 * D], rows_per_batch=St, **kw_t), after)
        ops.gemm_grouped(probs)
        for f in after:
            f()
        d2 = None
        if blk.dual:
            # attn2: self-attention over the image tokens only, input = LN(img) modulated by chunks 7 / 8 (shift_msa2, scale_msa2), residual gated by chunk 9
            # onto the stream AFTER the joint-attention residual (sd3/transformer.py:190-197)
            mi9 = mod[:, blk.mod_off:blk.mod_off + 9 * D]                        # (dual blocks never run tokenwise: refused in _engine_forward)
            Sip = (Si + 63) // 64 * 64
            n2a = ops.ln_modulate_fwd(img, mi9[:, 7 * D:8 * D], mi9[:, 6 * D:7 * D], Si)
            T2 = ops.gemm(n2a, blk.qkv2.lora.A_cat) if blk.qkv2.lora is not None else None
            kwq = dict(a2=T2, b2=blk.qkv2.lora.B_blk) if T2 is not None else {}
            qkv2 = ops.gemm(n2a, blk.qkv2.w, bias=blk.qkv2.b, **kwq)
            Q2 = torch.empty(B, H, Si, hd, dtype=BF16, device=dev); K2 = torch.empty_like(Q2)
            mk2 = torch.zeros if Sip > Si else torch.empty
            Q2t = K2t = None
            if not ops.ATTN_TR:
                Q2t = mk2(B, H, hd, Sip, dtype=BF16, device=dev); K2t = mk2(B, H, hd, Sip, dtype=BF16, device=dev)
            V2t = mk2(B, H, hd, Sip, dtype=BF16, device=dev)
            ops.qk_norm_rope_fwd(qkv2, blk.norm_q2, blk.norm_k2, cos, sin, Q2, K2, Q2t, K2t, V2t, B, H, hd, Si, 0, Si, Sip)
            O2 = torch.empty(B * Si, D, dtype=BF16, device=dev); lse2b = torch.empty(B, H, Si, dtype=F32, device=dev)
            ops.attn_fwd(Q2, K2, V2t, O2, lse2b, B, H, Si, Sip, hd, scale)
            del V2t
            ya2 = ybuf(B * Si)
            kw2 = dict(aux_out=ya2) if ya2 is not None else {}
            T_o2 = None
            if blk.to_out2.lora is not None:
                T_o2 = ops.gemm(O2, blk.to_out2.lora.A_cat)
                kw2.update(a2=T_o2, b2=blk.to_out2.lora.B_blk)
            x1b_img = ops.gemm(O2, blk.to_out2.w, bias=blk.to_out2.b, epilogue=EPI_GATE_RESIDUAL, aux_in=x1_img, gate=mi9[:, 8 * D:9 * D], rows_per_batch=Si, **kw2)
            d2 = SimpleNamespace(n2a=n2a, qkv2=qkv2, T2=T2, Q2=Q2, K2=K2, Q2t=Q2t, K2t=K2t, O2=O2, lse2b=lse2b, ya2=ya2, T_o2=T_o2, Sip=Sip)
            x1_img = x1b_img                       # what the MLP branch (and its residual) sees
        n2_i = ops.ln_modulate_fwd(x1_img, mi[:, 4 * D:5 * D], mi[:, 3 * D:4 * D], rpb)
        hpre_img = torch.empty(B * Si, 4 * D, dtype=BF16, device=dev)
        hpre_txt = x2_txt = n2_t = h_t = None
        yf_i, yf_t = ybuf(B * Si), (None if blk.last else ybuf(B * St))
        kf_i = dict(aux_out=yf_i) if yf_i is not None else {}
        kf_t = dict(aux_out=yf_t) if yf_t is not None else {}
        if blk.last:
            h_i = ops.gemm(n2_i, blk.ff1.w, bias=blk.ff1.b, epilogue=EPI_GELU, aux_out=hpre_img)
            x2_img = ops.gemm(h_i, blk.ff2.w, bias=blk.ff2.b, epilogue=EPI_GATE_RESIDUAL, aux_in=x1_img, gate=mi[:, 5 * D:6 * D],
                              rows_per_batch=rpb, **kf_i)
        else:
            n2_t = ops.ln_modulate_fwd(x1_txt, mt[:, 4 * D:5 * D], mt[:, 3 * D:4 * D], St)
            hpre_txt = torch.empty(B * St, 4 * D, dtype=BF16, device=dev)
            h_i, h_t = ops.gemm_grouped([dict(a=n2_i, w=blk.ff1.w, bias=blk.ff1.b, epilogue=EPI_GELU, aux_out=hpre_img),
                                         dict(a=n2_t, w=blk.ffc1.w, bias=blk.ffc1.b, epilogue=EPI_GELU, aux_out=hpre_txt)])
            x2_img, x2_txt = ops.gemm_grouped([
                dict(a=h_i, w=blk.ff2.w, bias=blk.ff2.b, epilogue=EPI_GATE_RESIDUAL, aux_in=x1_img, gate=mi[:, 5 * D:6 * D], rows_per_batch=rpb, **kf_i),
                dict(a=h_t, w=blk.ffc2.w, bias=blk.ffc2.b, epilogue=EPI_GATE_RESIDUAL, aux_in=x1_txt, gate=mt[:, 5 * D:6 * D], rows_per_batch=St, **kf_t)])
        if save:
            sv = SimpleNamespace(img=img, txt=txt, n_img=n_img, n_txt=n_txt if (T_txt is not None or full) else None, qkv=qkv, Q=Q, K=K,
                                 Qt=Qt, Kt=Kt, O=O, lse2=lse2, x1_img=x1_img, x1_txt=x1_txt, hpre_img=hpre_img,
                                 hpre_txt=hpre_txt, T_img=T_img, T_txt=T_txt, T_o=T_o, T_ao=T_ao, d2=d2)
            if full:    # a full fine-tune also needs every Linear's input (weight gradients) and the un-gated branch outputs (gate gradients)
                sv.n2_i, sv.n2_t, sv.h_i, sv.h_t, sv.ya_i, sv.ya_t, sv.yf_i, sv.yf_t = n2_i, n2_t, h_i, h_t, ya_i, ya_t, yf_i, yf_t
            return x2_img, x2_txt, sv
        return x2_img, x2_txt, None

    @staticmethod
    def _lk(lin):
        """(K2, k2_real, A_cat, B_blk, A_cat_T, B_blk_T) of a Linear's adapter group, zeros / None without one"""
        g = lin.lora if lin is not None else None
        if g is None:
            return 0, 0, None, None, None, None
        return g.K2, getattr(g, "k2_real", 0), g.A_cat, g.B_blk, g.A_cat_T, g.B_blk_T

    def _block_fwd_c(self, blk, img, txt, env, save: bool, ybuf):
        """_block_fwd through st355_block_sd3_joint_fwd: ONE C call sequences the launches of the host-side form below on the same operands —
        every buffer is allocated here and the kept ones go to the backward as before; bit-identical to it (ST355_BLOCK_ABI=0 restores the host sequencing)."""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        B, Si, St, S, Sp, mod, full = env.B, env.Si, env.St, env.S, env.Sp, env.mod, env.full
        e = lambda *sh, dt=BF16: torch.empty(*sh, dtype=dt, device=dev)
        last = blk.last
        K2q, kr_q, A_q, Bb_q, _, _ = self._lk(blk.qkv)
        K2a, kr_a, A_a, Bb_a, _, _ = self._lk(blk.add_qkv)
        K2o, kr_o, A_o, Bb_o, _, _ = self._lk(blk.to_out)
        K2t, kr_t, A_t, Bb_t, _, _ = self._lk(None if last else blk.to_add_out)
        et = (lambda r, c: _pad64_empty(r, c, dev)) if (full and save) else e        # text-stream tensors a full fine-tune contracts over tokens: zero-tailed to 64 rows
        n_img, n_txt, qkv, O, x1_img, hpre_img, n2_i, h_i, x2_img = (e(B * Si, D), et(B * St, D), e(B * S, 3 * D), e(B * S, D), e(B * Si, D), e(B * Si, 4 * D),
                                                                     e(B * Si, D), e(B * Si, 4 * D), e(B * Si, D))
        Q, K, lse2 = e(B, H, S, hd), e(B, H, S, hd), e(B, H, S, dt=F32)
        Vt = e(B, H, hd, Sp)
        if Sp > S:
            Vt[..., S:].zero_()                                 # only the pad columns need the zeros (the rest is written by the re-layout pass)
        x1_txt = hpre_txt = n2_t = h_t = x2_txt = None
        if not last:
            x1_txt, hpre_txt, n2_t, h_t, x2_txt = e(B * St, D), e(B * St, 4 * D), et(B * St, D), et(B * St, 4 * D), e(B * St, D)
        T_img = e(B * Si, K2q) if K2q else None
        T_txt = e(B * St, K2a) if K2a else None
        T_o = e(B * Si, K2o) if K2o else None
        T_ao = e(B * St, K2t) if K2t else None
        ya_i, ya_t, yf_i, yf_t = ybuf(B * Si), (None if last else ybuf(B * St)), ybuf(B * Si), (None if last else ybuf(B * St))
        c_img = e(B * Si, 3 * D) if (B > 1 and Si % 256) else None
        c_txt = e(B * St, 3 * D) if (B > 1 and St % 256) else None
        ops.block_sd3_joint_fwd(
            B=B, Si=Si, St=St, H=H, D=D, hd=hd, last=int(last), K2_qkv=K2q, k2r_qkv=kr_q, K2_aqkv=K2a, k2r_aqkv=kr_a, K2_out=K2o, k2r_out=kr_o, K2_aout=K2t, k2r_aout=kr_t,
            scale=env.scale, img=img, txt=txt, mod_img=mod[:, blk.mod_off:], mod_txt=mod[:, blk.mod_off_c:], mod_stride=mod.stride(0),
            w_qkv=blk.qkv.w, b_qkv=blk.qkv.b, w_add_qkv=blk.add_qkv.w, b_add_qkv=blk.add_qkv.b, w_out=blk.to_out.w, b_out=blk.to_out.b,
            w_add_out=None if last else blk.to_add_out.w, b_add_out=None if last else blk.to_add_out.b,
            w_ff1=blk.ff1.w, b_ff1=blk.ff1.b, w_ff2=blk.ff2.w, b_ff2=blk.ff2.b,
            w_ffc1=None if last else blk.ffc1.w, b_ffc1=None if last else blk.ffc1.b, w_ffc2=None if last else blk.ffc2.w, b_ffc2=None if last else blk.ffc2.b,
            A_qkv=A_q, Bb_qkv=Bb_q, A_aqkv=A_a, Bb_aqkv=Bb_a, A_out=A_o, Bb_out=Bb_o, A_aout=A_t, Bb_aout=Bb_t,
            norm_q=blk.norm_q, norm_k=blk.norm_k, norm_added_q=blk.norm_added_q, norm_added_k=blk.norm_added_k, cos=env.cos, sin=env.sin,
            n_img=n_img, n_txt=n_txt, qkv=qkv, Q=Q, K=K, O=O, lse2=lse2, x1_img=x1_img, x1_txt=x1_txt, hpre_img=hpre_img, hpre_txt=hpre_txt,
            T_img=T_img, T_txt=T_txt, T_o=T_o, T_ao=T_ao, ya_img=ya_i, ya_txt=ya_t, yf_img=yf_i, yf_txt=yf_t,
            n2_img=n2_i, n2_txt=n2_t, h_img=h_i, h_txt=h_t, Vt=Vt, c_img=c_img, c_txt=c_txt, out_img=x2_img, out_txt=x2_txt)
        if not save:
            return x2_img, x2_txt, None
        sv = SimpleNamespace(img=img, txt=txt, n_img=n_img, n_txt=n_txt if (T_txt is not None or full) else None, qkv=qkv, Q=Q, K=K, Qt=None, Kt=None, O=O, lse2=lse2,
                             x1_img=x1_img, x1_txt=x1_txt, hpre_img=hpre_img, hpre_txt=hpre_txt, T_img=T_img, T_txt=T_txt, T_o=T_o, T_ao=T_ao, d2=None)
        if full:
            sv.n2_i, sv.n2_t, sv.h_i, sv.h_t, sv.ya_i, sv.ya_t, sv.yf_i, sv.yf_t = n2_i, n2_t, h_i, h_t, ya_i, ya_t, yf_i, yf_t
        return x2_img, x2_txt, sv

    def _block_bwd_c(self, blk, sv, env_li, mod, cos, sin, d_img, d_txt, need_input_grads: bool, dmod=None):
        """the data path of one block's backward through st355_block_sd3_joint_bwd: returns (d_img', d_txt', G) with every intermediate gradient in G — the
        adapter gradients (LoRA) or the weight / bias / modulation gradients (full fine-tune) are taken from them by the caller, the same launches on the same
        operands as the host-side form, after the data path instead of between its steps (independent of it: bit-identical results)"""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        B, Si, St, S, Sp = env_li.B, env_li.Si, env_li.St, env_li.S, env_li.Sp
        e = lambda *sh: torch.empty(*sh, dtype=BF16, device=dev)
        last = blk.last
        K2q, kr_q, _, _, At_q, Bbt_q = self._lk(blk.qkv)
        K2a, kr_a, _, _, At_a, Bbt_a = self._lk(blk.add_qkv)
        K2o, kr_o, _, _, At_o, Bbt_o = self._lk(blk.to_out)
        K2t, kr_t, _, _, At_t, Bbt_t = self._lk(None if last else blk.to_add_out)
        G = SimpleNamespace(g_i=e(B * Si, D), dh_i=e(B * Si, 4 * D), dn2_i=e(B * Si, D), dx1_i=e(B * Si, D), dx1g_i=e(B * Si, D),
                            g_t=None, dh_t=None, dn2_t=None, dx1_t=None, dx1g_t=None,
                            dO=(torch.zeros if last else torch.empty)(B * S, D, dtype=BF16, device=dev), dqkv=e(B * S, 3 * D),
                            U_o=e(B * Si, K2o) if K2o else None, U_ao=e(B * St, K2t) if K2t else None, U_q=e(B * Si, K2q) if K2q else None,
                            U_a=e(B * St, K2a) if K2a else None, dn_i=None, dn_t=None, c_img=None, c_txt=None)
        et = (lambda r, c: _pad64_empty(r, c, dev)) if dmod is not None else e       # (full fine-tune: the text-stream gradients are weight-gradient operands)
        if not last:
            G.g_t, G.dh_t, G.dn2_t, G.dx1_t, G.dx1g_t = et(B * St, D), et(B * St, 4 * D), e(B * St, D), e(B * St, D), et(B * St, D)
        if B > 1 and Si % 256:
            G.c_img = e(B * Si, 3 * D)
        if B > 1 and St % 256:
            G.c_txt = et(B * St, 3 * D)
        d_img_out = d_txt_out = None
        if need_input_grads:
            G.dn_i, G.dn_t, d_img_out, d_txt_out = e(B * Si, D), e(B * St, D), e(B * Si, D), e(B * St, D)
        dQ, dK = e(B, H, S, hd), e(B, H, S, hd)
        st = {}
        if dmod is not None:
            # full fine-tune: the block's modulation / gate / bias gradients ride in the entry's own passes (csrc/stats.hip) — dmod = the fp32 [B, mod_total] buffer
            st = dict(dmod_img=dmod[:, blk.mod_off:], dmod_txt=dmod[:, blk.mod_off_c:], dmod_stride=dmod.stride(0), ya_img=sv.ya_i, ya_txt=sv.ya_t, yf_img=sv.yf_i,
                      yf_txt=sv.yf_t, gb_ff2=blk.ff2.gb, gb_ff1=blk.ff1.gb, gb_out=blk.to_out.gb, gb_qkv=blk.qkv.gb, gb_add_qkv=blk.add_qkv.gb,
                      gb_ffc2=None if last else blk.ffc2.gb, gb_ffc1=None if last else blk.ffc1.gb, gb_add_out=None if last else blk.to_add_out.gb)
        ops.block_sd3_joint_bwd(**st,
            B=B, Si=Si, St=St, H=H, D=D, hd=hd, last=int(last), need_input_grads=int(need_input_grads),
            K2_qkv=K2q, k2r_qkv=kr_q, K2_aqkv=K2a, k2r_aqkv=kr_a, K2_out=K2o, k2r_out=kr_o, K2_aout=K2t, k2r_aout=kr_t, scale=1.0 / math.sqrt(hd),
            img=sv.img, txt=sv.txt, mod_img=mod[:, blk.mod_off:], mod_txt=mod[:, blk.mod_off_c:], mod_stride=mod.stride(0),
            qkv=sv.qkv, Q=sv.Q, K=sv.K, O=sv.O, lse2=sv.lse2, x1_img=sv.x1_img, x1_txt=sv.x1_txt, hpre_img=sv.hpre_img, hpre_txt=sv.hpre_txt,
            wT_qkv=blk.qkv.wT, wT_add_qkv=blk.add_qkv.wT, wT_out=blk.to_out.wT, wT_add_out=None if last else blk.to_add_out.wT,
            wT_ff1=blk.ff1.wT, wT_ff2=blk.ff2.wT, wT_ffc1=None if last else blk.ffc1.wT, wT_ffc2=None if last else blk.ffc2.wT,
            At_qkv=At_q, Bbt_qkv=Bbt_q, At_aqkv=At_a, Bbt_aqkv=Bbt_a, At_out=At_o, Bbt_out=Bbt_o, At_aout=At_t, Bbt_aout=Bbt_t,
            norm_q=blk.norm_q, norm_k=blk.norm_k, norm_added_q=blk.norm_added_q, norm_added_k=blk.norm_added_k, cos=cos, sin=sin,
            d_img=d_img, d_txt=d_txt, g_img=G.g_i, g_txt=G.g_t, dh_img=G.dh_i, dh_txt=G.dh_t, dn2_img=G.dn2_i, dn2_txt=G.dn2_t, dx1_img=G.dx1_i, dx1g_img=G.dx1g_i,
            dx1_txt=G.dx1_t, dx1g_txt=G.dx1g_t, U_o=G.U_o, U_ao=G.U_ao, dO=G.dO, dqkv=G.dqkv, dQ=dQ, dK=dK, U_qkv=G.U_q, U_aqkv=G.U_a,
            dn_img=G.dn_i, dn_txt=G.dn_t, c_img=G.c_img, c_txt=G.c_txt, d_img_out=d_img_out, d_txt_out=d_txt_out)
        # each stream's rows of dqkv in the operand form the C entry used: in place (a 3-D view, segmented) when B == 1 or tile-aligned, else its compact copy
        envs = SimpleNamespace(B=B, S=S)
        G.dq_i = G.c_img if G.c_img is not None else _FluxEngine._compact(_rows3(G.dqkv, 0, Si, B, S), envs, Si)
        G.dq_t = G.c_txt if G.c_txt is not None else _FluxEngine._compact(_rows3(G.dqkv, Si, St, B, S), envs, St)
        return d_img_out, d_txt_out, G

    def _engine_forward(self, latents, enc, pooled, timestep, save: bool, full: bool = False):
        D, H, hd = self.D, self.H, self.hd
        B, C, Hh, Ww = latents.shape
        h, w = Hh // 2, Ww // 2
        Si, St = h * w, enc.shape[1]
        S = Si + St
        Sp = (S + 63) // 64 * 64
        dev = self.device_
        for g in self.lora_groups:
            g.pack()
        pos, cos, sin = self._tables(h, w, S, B)
        # ---- embeddings (sd3/transformer.py:623-700) ----
        patches = ops.patchify(latents, order=0).view(B * Si, 4 * C)
        img = ops.gemm(patches, self.l_patch.w, bias=self.l_patch.b, epilogue=EPI_ADD, aux_in=pos)
        enc2d = enc.reshape(B * St, -1).contiguous()
        txt = ops.gemm(enc2d, self.l_ctx.w, bias=self.l_ctx.b)
        tokenwise = timestep.dim() == 2
        if tokenwise and tuple(timestep.shape) != (B, Si):
            raise ValueError(f"SD3 tokenwise timesteps expected sequence length {Si}, got {timestep.shape[1]}.")      # sd3/transformer.py:625-626
        t32 = timestep.to(device=dev, dtype=F32).reshape(-1).contiguous()
        tproj = ops.timestep_proj(t32, 256, 1.0)
        t1 = ops.gemm(tproj, self.l_t1.w, bias=self.l_t1.b); st1 = ops.silu(t1)
        pooled_b = pooled.to(BF16).contiguous()
        p1 = ops.gemm(pooled_b, self.l_p1.w, bias=self.l_p1.b); sp1 = ops.silu(p1)
        mod_i, rpb_i = None, Si
        if tokenwise:
            # TOKENWISE timesteps [B, S_img] (CREPA self-flow; sd3/transformer.py:61-75, 126-142, 680-685, 876): one conditioning row per image token — the image
            # stream's shift / scale / gate rows and norm_out's (scale, shift) are per TOKEN (the AdaLN / gated-residual kernels index their modulation row by
            # row // rows_per_batch: rows_per_batch = 1), the context stream is conditioned on the mean over the tokens.  The per-token modulation rows of every
            # block are ONE GEMM [B * S_img, D] x [mod_total, D]^T (B * S_img * mod_total bf16: 14 GB for SD3-Medium at 1024^2, batch 8 — HBM holds it).
            if full:
                raise NotImplementedError("tokenwise timesteps under a full fine-tune (per-token modulation gradients) are not implemented on the st355 path")
            if any(b.dual for b in self.blocks):
                raise NotImplementedError("tokenwise timesteps with SD3.5 dual-attention blocks: the reference chunks SD35AdaLayerNormZeroX's [B, S, 9D] output along the "
                                          "token axis (sd3/transformer.py:155-164) — not a defined computation")
            pe = ops.gemm(sp1, self.l_p2.w, bias=self.l_p2.b)                                         # [B, D]: the pooled-text embedding, shared by a sample's tokens
            temb_tok = ops.gemm(st1, self.l_t2.w, bias=self.l_t2.b, epilogue=EPI_ADD, aux_in=pe[:, None, :].expand(B, Si, D).reshape(B * Si, D))
            tsum = torch.empty(B, D, dtype=F32, device=dev)
            ops.colsum_prod(temb_tok, tsum, rows_per_batch=Si)
            temb = (tsum / Si).to(BF16)                                                               # temb_context = temb.mean(dim=1)   (:681-682)
            mod_i = ops.gemm(ops.silu(temb_tok), self.mod_w, bias=self.mod_b)                         # [B * S_img, mod_total]
            rpb_i = 1
        else:
            temb = ops.add(ops.gemm(st1, self.l_t2.w, bias=self.l_t2.b), ops.gemm(sp1, self.l_p2.w, bias=self.l_p2.b))
        st = ops.silu(temb)
        mod = ops.gemm(st, self.mod_w, bias=self.mod_b)
        if mod_i is None:
            mod_i = mod
        ctx = SimpleNamespace(B=B, Si=Si, St=St, S=S, Sp=Sp, cos=cos, sin=sin, mod=mod, mod_i=mod_i, rpb_i=rpb_i, blocks=[], C=C, Hh=Hh, Ww=Ww)
        if full and save:
            ctx.emb = SimpleNamespace(patches=patches, enc2d=enc2d, tproj=tproj, t1=t1, st1=st1, pooled=pooled_b, p1=p1, sp1=sp1, temb=temb, st=st)
        ybuf = (lambda rows: torch.empty(rows, D, dtype=BF16, device=dev)) if (full and save) else (lambda rows: None)
        scale = 1.0 / math.sqrt(hd)
        env = SimpleNamespace(B=B, Si=Si, St=St, S=S, Sp=Sp, cos=cos, sin=sin, mod=mod, mod_i=mod_i, rpb_i=rpb_i, scale=scale, full=full)
        ctx.env, ctx.blocks, ctx.ck = env, [None] * len(self.blocks), {}
        # activation-checkpoint plan (sd3/transformer.py:716-833; training/checkpoint_plan.py): a checkpointed segment keeps only its input
        from ..training.checkpoint_plan import segments as _segments
        ctx.segs = _segments(len(self.blocks), bool(save and self.gradient_checkpointing), self.gradient_checkpointing_interval,
                             self.gradient_checkpointing_segment_stride)
        # TREAD routing (sd3/transformer.py:694-706, 796-803; training/tread.py): only while training, between the route's two blocks the IMAGE stream is a
        # per-sample subset of its tokens.  Under routing the reference drops the segmented checkpoint form for the per-block one (:716-728)
        from ..training.tread import normalise_routes
        routes = normalise_routes(self._tread_routes, len(self.blocks)) if (save and self.training and self._tread_router is not None) else []
        if routes and tokenwise:
            raise NotImplementedError("tokenwise timesteps under TREAD routing (the routed tokens' modulation rows would be gathered too) are not implemented")
        if routes:
            from ..training.checkpoint_plan import per_block as _per_block
            ctx.segs = _per_block(len(self.blocks), bool(self.gradient_checkpointing), self.gradient_checkpointing_interval, self.gradient_checkpointing_segment_stride)
        ctx.envs, ctx.route_start, ctx.route_end = [env] * len(self.blocks), {}, {}
        rp, info, saved, env_cur = 0, None, None, env
        for (s0, n, ck) in ctx.segs:
            for bi in range(s0, s0 + n):
                if rp < len(routes) and info is None and bi == routes[rp]["start_layer_idx"]:
                    info = self._tread_router.get_mask(img.view(B, Si, D), mask_ratio=routes[rp]["selection_ratio"], force_keep=getattr(self, "_force_keep_mask", None))
                    saved = img
                    img = ops.gather_rows(img.view(B, Si, D), info.keep_i32()).view(-1, D)          # TREADRouter.start_route
                    K = info.ids_keep.shape[1]
                    env_cur = SimpleNamespace(**{**vars(env), "Si": K, "S": K + St, "Sp": (K + St + 63) // 64 * 64})
                    ctx.route_start[bi] = info
                if ck and bi == s0:
                    ctx.ck[s0] = (img, txt)
                ctx.envs[bi] = env_cur
                img, txt, ctx.blocks[bi] = self._block_fwd(self.blocks[bi], img, txt, env_cur, save and not ck)
                if info is not None and bi == routes[rp]["end_layer_idx"]:
                    full_seq = saved.clone()                                                        # TREADRouter.end_route(original_x=saved)
                    ops.scatter_rows(img.view(B, env_cur.Si, D), info.keep_i32(), full_seq.view(B, Si, D))
                    img, ctx.route_end[bi] = full_seq, info
                    info, saved, env_cur, rp = None, None, env, rp + 1
        if info is not None:
            raise ValueError("TREAD route does not end inside the block stack (end_layer_idx)")
        # ---- output head: AdaLayerNormContinuous (scale, shift), proj_out, unpatchify "nhwpqc->nchpwq" ----
        mo = mod_i[:, self.mod_off_out:self.mod_off_out + 2 * D]
        n_out = ops.ln_modulate_fwd(img, mo[:, :D], mo[:, D:2 * D], rpb_i)
        out = ops.gemm(n_out, self.l_out.w, bias=self.l_out.b)
        if save:
            ctx.x_img_final = img
            if full:
                ctx.n_out = n_out
        return ops.unpatchify(out.view(B, Si, -1), self.out_channels, Hh, Ww, order=1), ctx

    def set_router(self, router, routes):
        """sd3/transformer.py:407-409: TREAD router + [{selection_ratio, start_layer_idx, end_layer_idx}] (training/tread.py)"""
        self._tread_router, self._tread_routes = router, routes

    def _recompute_segment(self, ctx, li: int):
        """backward reached block `li` of a checkpointed segment: re-run the segment's forward from its kept input, this time keeping the activations"""
        s0, n = next((a, c) for (a, c, ck) in ctx.segs if ck and a <= li < a + c)
        img, txt = ctx.ck.pop(s0)
        for bi in range(s0, s0 + n):
            img, txt, ctx.blocks[bi] = self._block_fwd(self.blocks[bi], img, txt, ctx.envs[bi], True)

    def _engine_backward(self, ctx, dout):
        if not self._prepared:
            raise RuntimeError("call prepare_for_training() after loading weights (builds the K-major dgrad operands)")
        D, H, hd = self.D, self.H, self.hd
        B, Si, St, S, Sp, mod, cos, sin = ctx.B, ctx.Si, ctx.St, ctx.S, ctx.Sp, ctx.mod, ctx.cos, ctx.sin
        dev = self.device_
        scale = 1.0 / math.sqrt(hd)
        dpk = ops.patchify(dout.to(BF16).contiguous(), order=1).view(B * Si, -1)
        mod_i, rpb = ctx.mod_i, ctx.rpb_i          # the image stream's modulation rows: per sample, or per token under tokenwise timesteps (rows_per_batch = 1)
        mo = mod_i[:, self.mod_off_out:self.mod_off_out + 2 * D]
        dn = ops.gemm(dpk, self.l_out.wT)
        d_img, _ = ops.ln_modulate_bwd(dn, ctx.x_img_final, mo[:, :D], rpb)
        d_txt = None
        del dn, dpk
        for li in range(len(self.blocks) - 1, -1, -1):
            if ctx.blocks[li] is None:
                self._recompute_segment(ctx, li)
            if li in ctx.route_end:                       # backward enters a TREAD route at its END: the routed blocks see only the kept tokens' gradient rows
                r_info = ctx.route_end[li]
                d_full = d_img
                d_img = ops.gather_rows(d_full.view(B, ctx.Si, D), r_info.keep_i32()).view(-1, D)
            Si, S, Sp = ctx.envs[li].Si, ctx.envs[li].S, ctx.envs[li].Sp
            blk, sv = self.blocks[li], ctx.blocks[li]
            ctx.blocks[li] = None
            mi = mod_i[:, blk.mod_off:blk.mod_off + 6 * D]
            mt = mod[:, blk.mod_off_c:blk.mod_off_c + (2 if blk.last else 6) * D]
            rpb_i = rpb if rpb == 1 else Si             # (inside a TREAD route Si is the kept-token count)
            if _BLOCK_ABI and not blk.dual and ops.ATTN_TR and sv.Qt is None and d_img.is_contiguous() and (d_txt is None or d_txt.is_contiguous()) and rpb != 1:
                # the data path as ONE C entry point (st355_block_sd3_joint_bwd), then the rank-space adapter gradients from the gradients it left behind
                envs = SimpleNamespace(B=B, S=S)
                d_img, d_txt, G = self._block_bwd_c(blk, sv, ctx.envs[li], mod, cos, sin, d_img, d_txt, li != 0)
                pairs = [(blk.to_out, G.U_o, sv.T_o, G.dx1g_i, 0, Si)]
                if not blk.last:
                    pairs.append((blk.to_add_out, G.U_ao, sv.T_ao, G.dx1g_t, Si, St))
                for (lin, U, T_, dxg, lo, rows) in pairs:
                    if lin.lora is not None:
                        lin.lora.grads(_FluxEngine._compact(_rows3(sv.O, lo, rows, B, S), envs, rows), T_, dxg, U, self.accumulate_lora_grads, self.grad_sync)
                for (lin, dq, n_in, T_, U) in ((blk.qkv, G.dq_i, sv.n_img, sv.T_img, G.U_q), (blk.add_qkv, G.dq_t, sv.n_txt, sv.T_txt, G.U_a)):
                    if lin.lora is not None:
                        lin.lora.grads(n_in, T_, dq, U, self.accumulate_lora_grads, self.grad_sync)
                del sv, G
                if li in ctx.route_start and d_img is not None:
                    ops.scatter_rows(d_img.view(B, Si, D), ctx.route_start[li].keep_i32(), d_full.view(B, ctx.Si, D))
                    d_img, d_full = d_full, None
                continue
            g_i = ops.scale_cols(d_img, mi[:, 5 * D:6 * D], rpb_i)
            if blk.last:
                dh_i = ops.gemm(g_i, blk.ff2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_img)
                dn2_i = ops.gemm(dh_i, blk.ff1.wT)
                dx1_t = dx1g_t = None
            else:
                g_t = ops.scale_cols(d_txt, mt[:, 5 * D:6 * D], St)
                dh_i, dh_t = ops.gemm_grouped([dict(a=g_i, w=blk.ff2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_img),
                                               dict(a=g_t, w=blk.ffc2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_txt)])
                dn2_i, dn2_t = ops.gemm_grouped([dict(a=dh_i, w=blk.ff1.wT), dict(a=dh_t, w=blk.ffc1.wT)])
                dx1_t, dx1g_t = ops.ln_modulate_bwd(dn2_t, sv.x1_txt, mt[:, 4 * D:5 * D], St, dres=d_txt, gate=mt[:, 2 * D:3 * D], want_gated=True)
                del g_t, dh_t, dn2_t
            dn2a = None
            if blk.dual:
                # the MLP's LayerNorm read the stream AFTER the attn2 residual: its gated gradient feeds attn2's output projection, the un-gated one is the
                # gradient of the stream after the joint-attention residual (dx1) — attn2 first, then the joint attention as for a regular block
                d2, mi9 = sv.d2, mod[:, blk.mod_off:blk.mod_off + 9 * D]
                dx1_i, dxg2 = ops.ln_modulate_bwd(dn2_i, sv.x1_img, mi[:, 4 * D:5 * D], Si, dres=d_img, gate=mi9[:, 8 * D:9 * D], want_gated=True)
                lo2, lq2 = blk.to_out2.lora, blk.qkv2.lora
                U2 = ops.gemm(dxg2, lo2.B_blk_T) if lo2 is not None else None
                dO2 = ops.gemm(dxg2, blk.to_out2.wT, **(dict(a2=U2, b2=lo2.A_cat_T) if U2 is not None else {}))
                if lo2 is not None:
                    lo2.grads(d2.O2, d2.T_o2, dxg2, U2, self.accumulate_lora_grads, self.grad_sync)
                dqkv2 = torch.empty(B * Si, 3 * D, dtype=BF16, device=dev)
                dQ2 = torch.empty(B, H, Si, hd, dtype=BF16, device=dev); dK2 = torch.empty_like(dQ2)
                ops.attn_bwd(d2.Q2, d2.K2, d2.Q2t, d2.K2t, d2.qkv2[:, 2 * D:], d2.O2, dO2, d2.lse2b, dQ2, dK2, dqkv2[:, 2 * D:], B, H, Si, d2.Sip, hd, scale)
                ops.qk_norm_rope_bwd(dQ2, dK2, d2.qkv2, blk.norm_q2, blk.norm_k2, cos, sin, dqkv2, B, H, hd, Si, 0, Si)
                Uq2 = ops.gemm(dqkv2, lq2.B_blk_T) if lq2 is not None else None
                if li != 0:
                    dn2a = ops.gemm(dqkv2, blk.qkv2.wT, **(dict(a2=Uq2, b2=lq2.A_cat_T) if Uq2 is not None else {}))
                if lq2 is not None:
                    lq2.grads(d2.n2a, d2.T2, dqkv2, Uq2, self.accumulate_lora_grads, self.grad_sync)
                dx1g_i = ops.scale_cols(dx1_i, mi[:, 2 * D:3 * D], Si)
                del dxg2, dO2, dQ2, dK2, dqkv2, U2, Uq2, d2
            else:
                dx1_i, dx1g_i = ops.ln_modulate_bwd(dn2_i, sv.x1_img, mi[:, 4 * D:5 * D], rpb_i, dres=d_img, gate=mi[:, 2 * D:3 * D], want_gated=True)
            del g_i, dh_i, dn2_i
            # attention output projections -> dO rows of both streams (+ adapter grads); a context_pre_only block has no txt rows
            dO = (torch.zeros if blk.last else torch.empty)(B * S, D, dtype=BF16, device=dev)
            U_i = ops.gemm(dx1g_i, blk.to_out.lora.B_blk_T) if blk.to_out.lora is not None else None
            U_t = ops.gemm(dx1g_t, blk.to_add_out.lora.B_blk_T) if (not blk.last and blk.to_add_out.lora is not None) else None
            after = []
            kw_i = dict(a2=U_i, b2=blk.to_out.lora.A_cat_T) if U_i is not None else {}
            probs = _stream_problems(B, S, Si, dict(a=dx1g_i, w=blk.to_out.wT, out=_rows3(dO, 0, Si, B, S), **kw_i), after)
            if not blk.last:
                kw_t = dict(a2=U_t, b2=blk.to_add_out.lora.A_cat_T) if U_t is not None else {}
                probs += _stream_problems(B, S, St, dict(a=dx1g_t, w=blk.to_add_out.wT, out=_rows3(dO, Si, St, B, S), **kw_t), after)
            ops.gemm_grouped(probs)
            for f in after:
                f()
            pairs = [(blk.to_out, U_i, sv.T_o, dx1g_i, 0, Si)]
            if not blk.last:
                pairs.append((blk.to_add_out, U_t, sv.T_ao, dx1g_t, Si, St))
            envs = SimpleNamespace(B=B, S=S)
            for (lin, U, T_, dxg, lo, rows) in pairs:
                if lin.lora is not None:       # this stream's rows of the joint attention output: read in place when tile-aligned (image rows), else a compact copy
                    lin.lora.grads(_FluxEngine._compact(_rows3(sv.O, lo, rows, B, S), envs, rows), T_, dxg, U, self.accumulate_lora_grads, self.grad_sync)
            del dx1g_i, dx1g_t, U_i, U_t
            dqkv = torch.empty(B * S, 3 * D, dtype=BF16, device=dev)
            dQ = torch.empty(B, H, S, hd, dtype=BF16, device=dev); dK = torch.empty_like(dQ)
            ops.attn_bwd(sv.Q, sv.K, sv.Qt, sv.Kt, sv.qkv[:, 2 * D:], sv.O, dO, sv.lse2, dQ, dK, dqkv[:, 2 * D:], B, H, S, Sp, hd, scale)
            ops.qk_norm_rope_bwd(dQ, dK, sv.qkv, blk.norm_q, blk.norm_k, cos, sin, dqkv, B, H, hd, Si, 0, S)
            ops.qk_norm_rope_bwd(dQ, dK, sv.qkv, blk.norm_added_q, blk.norm_added_k, cos, sin, dqkv, B, H, hd, St, Si, S)
            del dQ, dK, dO
            first = li == 0
            # the two streams' rows of the joint dqkv: the image rows in place (segmented operands), the 154 text rows as a compact copy
            dq_i = _FluxEngine._compact(_rows3(dqkv, 0, Si, B, S), envs, Si); dq_t = _FluxEngine._compact(_rows3(dqkv, Si, St, B, S), envs, St)
            streams = [("img", blk.qkv, dq_i, sv.n_img, sv.T_img, Si), ("txt", blk.add_qkv, dq_t, sv.n_txt, sv.T_txt, St)]
            if first:
                streams = [s_ for s_ in streams if s_[1].lora is not None]     # frozen embedders: only adapter grads remain
            probs, Us, dns = [], {}, []
            for (name, lin, dq, n_in, T_, rows) in streams:
                kw = {}
                if lin.lora is not None:
                    Us[name] = torch.empty(B * rows, lin.lora.B_blk_T.shape[0], dtype=BF16, device=dev)
                    ops.gemm(dq, lin.lora.B_blk_T, out=Us[name])
                    kw = dict(a2=Us[name], b2=lin.lora.A_cat_T)
                if not first:
                    dns.append(torch.empty(B * rows, D, dtype=BF16, device=dev))
                    probs.append(dict(a=dq, w=lin.wT, out=dns[-1], **kw))
            if probs:
                ops.gemm_grouped(probs)
            for (name, lin, dq, n_in, T_, rows) in streams:
                if lin.lora is not None:
                    lin.lora.grads(n_in, T_, dq, Us[name], self.accumulate_lora_grads, self.grad_sync)
            if not first:
                d_img, _ = ops.ln_modulate_bwd(dns[0], sv.img, mi[:, D:2 * D], rpb_i, dres=dx1_i)
                if dn2a is not None:             # the second reader of LN(img): attn2's modulated input (scale_msa2)
                    d_img, _ = ops.ln_modulate_bwd(dn2a, sv.img, mod[:, blk.mod_off + 7 * D:blk.mod_off + 8 * D], Si, dres=d_img)
                c_scale = mt[:, :D] if blk.last else mt[:, D:2 * D]
                d_txt, _ = ops.ln_modulate_bwd(dns[1], sv.txt, c_scale, St, dres=dx1_t)
            del dqkv, sv, dns
            if li in ctx.route_start and d_img is not None:   # ... and leaves it at its START: skipped tokens keep the gradient they had at the route's end
                ops.scatter_rows(d_img.view(B, Si, D), ctx.route_start[li].keep_i32(), d_full.view(B, ctx.Si, D))
                d_img, d_full = d_full, None
        return None

    # ------------------------------------------------------------------------------------------------
    # full fine-tune (BASELINE.json configs[3]): every weight, bias and modulation row trains
    # ------------------------------------------------------------------------------------------------
    def _all_linears(self):
        ls = [self.l_patch, self.l_t1, self.l_t2, self.l_p1, self.l_p2, self.l_ctx, self.l_out]
        for blk in self.blocks:
            ls += [l for l in (blk.qkv, blk.add_qkv, blk.to_out, blk.to_add_out, blk.qkv2, blk.to_out2, blk.ff1, blk.ff2, blk.ffc1, blk.ffc2) if l is not None]
        return ls

    def enable_full_finetune(self):
        """gradient arena with the weight arena's layout; every base parameter becomes trainable (bf16 params, bf16 grads)"""
        if self.lora_groups:
            raise RuntimeError("full fine-tune and LoRA adapters are exclusive")
        self.full = True
        # TWO gradient arenas (round 6): the backward fills one and hands it to autograd as it is; the next backward fills the other.  Round 5 cloned the arena at
        # the end of every backward so that `.grad` never aliased the buffer the next backward overwrites (4 GB read + written per SD3-Medium step); with two
        # arenas the hand-over is free, and under gradient accumulation autograd adds the new arena into the one `.grad` still holds (_pick_grad_arena)
        self._grad_arenas = [torch.zeros_like(self.arena), torch.zeros_like(self.arena)]
        self._grad_sel = 0
        self.grad_arena = self._grad_arenas[0]
        base = self.arena.data_ptr()
        self._grad_views = ([], [])          # per arena: (owner, attribute, view) of every gradient view the backward writes through

        def gview(owner, attr, t):
            off = (t.data_ptr() - base) // 2
            for k in (0, 1):
                self._grad_views[k].append((owner, attr, None if t is None else self._grad_arenas[k][off:off + t.numel()].view(t.shape)))
            setattr(owner, attr, self._grad_views[0][-1][2])

        for l in self._all_linears():
            gview(l, "gw", l.w); gview(l, "gb", l.b)
        for blk in self.blocks:          # q/k RMSNorm weights (SD3.5): gradient views like every other parameter
            for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k", "norm_q2", "norm_k2"):
                w = getattr(blk, nm, None)
                if w is not None:
                    gview(blk, "g_" + nm, w)
                else:
                    setattr(blk, "g_" + nm, None)
        gview(self, "g_mod_w", self.mod_w); gview(self, "g_mod_b", self.mod_b)
        ps = sorted([p for n, p in self.named_parameters() if ".lora_" not in n], key=lambda p: p.data_ptr())
        for p in ps:
            p.requires_grad_(True)
        self._full_params = ps
        self._full_offsets = [((p.data_ptr() - base) // 2, p.numel()) for p in ps]
        self.prepare_for_training()
        for l in (self.l_t2, self.l_p2):
            l.wT = l.w.t().contiguous()
        return ps

    def _pick_grad_arena(self):
        """before a full fine-tune backward: write into the arena `.grad` does NOT alias (after optimizer.zero_grad(set_to_none=True): either; in the middle of a
        gradient accumulation: the one autograd is not accumulating into)"""
        g = self._full_params[0].grad if self._full_params else None
        if g is not None:
            off0 = self._full_offsets[0][0] * 2
            if g.data_ptr() == self._grad_arenas[self._grad_sel].data_ptr() + off0:
                self._select_grad_arena(1 - self._grad_sel)

    def _select_grad_arena(self, k: int):
        if k == self._grad_sel:
            return
        self._grad_sel = k
        self.grad_arena = self._grad_arenas[k]
        for owner, attr, view in self._grad_views[k]:
            setattr(owner, attr, view)
        if self.grad_sync is not None:
            self.grad_sync.flat = self.grad_arena                     # the exchange walks the arena this backward fills (same offsets)

    def _swap_grad_arena(self):
        """training.grad_sync.hand_over_gradients: the arena just filled now belongs to autograd; the next backward takes the other one"""
        self._select_grad_arena(1 - self._grad_sel)

    def trainable_parameters(self):
        return list(self._full_params) if getattr(self, "full", False) else list(self._lora_params)

    def diffusers_state_dict(self) -> Dict[str, torch.Tensor]:
        """{diffusers checkpoint key: tensor} of the base parameters + the position table (the parameter names ARE the checkpoint keys): what `save_pretrained`
        writes for a full fine-tune (training/trainer.py save_state)"""
        sd = {k: v.detach() for k, v in self.named_parameters() if ".lora_" not in k}
        sd["pos_embed.pos_embed"] = self.pos_embed.pos_embed.detach()
        return sd

    def load_diffusers_state(self, state: Dict[str, torch.Tensor]):
        self.load_flat_state(state)

    def _refresh_transposed(self):
        """W^T follows the weights (2 B read + 2 B write per parameter; ~1 ms for SD3-Medium)"""
        for l in self._all_linears():
            if getattr(l, "wT", None) is not None:
                ops.transpose(l.w, out=l.wT)

    def _engine_backward_full(self, ctx, dout):
        D, H, hd = self.D, self.H, self.hd
        B, Si, St, S, Sp, mod, cos, sin = ctx.B, ctx.Si, ctx.St, ctx.S, ctx.Sp, ctx.mod, ctx.cos, ctx.sin
        dev = self.device_
        scale = 1.0 / math.sqrt(hd)
        self._refresh_transposed()
        dmod = torch.zeros(B, self.mod_total, dtype=F32, device=dev)        # d loss / d (modulation linear output)
        tmp_b = {}

        def pad64_rows(rows, cols):
            """an uninitialised [rows, cols] buffer that lives in a zero-tailed parent with a multiple of 64 rows: P64 hands the parent to the TN GEMM, no copy"""
            return _pad64_empty(rows, cols, dev)

        def P64(t):
            """the operand with a multiple of 64 contraction rows (the TN GEMM's granule): as is when aligned or segmented, its zero-tailed parent when it was
            allocated by _pad64_empty, else a zero-padded copy"""
            if t.dim() == 3:
                return t
            r = t.shape[0]
            if r % 64 == 0 and t.is_contiguous():
                return t
            par = getattr(t, "_st355_pad64", None)
            if par is not None:
                return par
            o = torch.zeros((r + 63) // 64 * 64, t.shape[1], dtype=BF16, device=dev)
            o[:r] = t
            return o

        def wgrad(lin, dy, x, bias: bool = True):
            """dW = dY^T X ; db = colsum(dY)   (into the gradient arena views of `lin`; bias=False: the bias gradient was already taken by a fused pass)"""
            ops.gemm_tn(P64(dy), P64(x), out=lin.gw)
            if not bias:
                return
            N = dy.shape[1]
            t = tmp_b.get(N)
            if t is None:
                t = tmp_b[N] = torch.empty(1, N, dtype=F32, device=dev)
            ops.colsum_prod(dy, t)
            lin.gb.copy_(t[0])

        def mod_grads(dn, x_in, rows, k_shift, k_scale, dm):
            """d shift = sum_t dY, d scale = sum_t dY * LN(x) of one AdaLN instance (chunk indices k_* inside its slice dm of the modulation gradient); x_in = the
            LayerNorm's input.  (LN(x) is recomputed: recovering it from the saved modulated output divides by 1 + scale, singular where a scale entry is -1.)"""
            ops.colsum_prod(dn, dm[:, k_shift * D:(k_shift + 1) * D], rows_per_batch=rows)
            ops.colsum_prod(dn, dm[:, k_scale * D:(k_scale + 1) * D], b=ops.layer_norm_xhat(x_in), rows_per_batch=rows)

        def qk_bwd(dQ_, dK_, qkv_, wq, wk, dqkv_, rows, pos0, S_, gq, gk):
            """RMSNorm (+ identity RoPE) backward of one stream's q / k; with norm weights (SD3.5) also their gradients (sd3/transformer.py:155-165)"""
            if wq is None and wk is None:
                ops.qk_norm_rope_bwd(dQ_, dK_, qkv_, wq, wk, cos, sin, dqkv_, B, H, hd, rows, pos0, S_)
            else:
                ops.qk_norm_rope_bwd_wgrad(dQ_, dK_, qkv_, wq, wk, cos, sin, dqkv_, B, H, hd, rows, pos0, S_, gq, gk)

        def rows_of(t, lo, n, seg: bool = False):
            """rows [lo, lo+n) of every batch element of a joint [B*S, C] buffer as a weight-gradient operand.  seg: in place — a [B, n, C] strided view, the
            segmented-contraction form of st355_gemm_tn_seg_bf16 — when n is a whole number of 64-row K-tiles (the image rows of every bucket); else (the text
            rows, and callers that also read the operand as a plain matrix) a compact copy, its row count zero-tailed to 64 (P64 below takes it as is)"""
            if B == 1:
                return t[lo:lo + n]
            v = t.view(B, S, -1)[:, lo:lo + n]
            if seg and n % 64 == 0 and n >= 128:
                return v
            o = pad64_rows(B * n, t.shape[1])
            o.view(B, n, -1).copy_(v)
            return o

        # ---- head ----
        dpk = ops.patchify(dout.to(BF16).contiguous(), order=1).view(B * Si, -1)
        mo = mod[:, self.mod_off_out:self.mod_off_out + 2 * D]
        dmo = dmod[:, self.mod_off_out:self.mod_off_out + 2 * D]
        wgrad(self.l_out, dpk, ctx.n_out)
        dn = ops.gemm(dpk, self.l_out.wT)
        mod_grads(dn, ctx.x_img_final, Si, 1, 0, dmo)                        # AdaLayerNormContinuous: (scale, shift)
        d_img, _ = ops.ln_modulate_bwd(dn, ctx.x_img_final, mo[:, :D], Si)
        d_txt = None
        del dn, dpk
        sync = self.grad_sync
        if sync is not None:
            sync.ready(self._head_arena_lo, self.grad_arena.numel())        # proj_out gradients are final
        # The fused modulation matrix (every block's adaLN Linear as rows of ONE [mod_total, D] matrix: a third of SD3-Medium's parameters, 1.35 GB of gradient) gets its
        # gradient rows block by block (r6) — dW_mod[r0:r1] = dmod[:, r0:r1]^T silu(temb) as soon as the block that owns rows [r0, r1) has run — so that the exchange
        # can take them behind the backward; as one product at the end of the backward the whole region left as one exposed 1.5 GB slice.  Same arithmetic per row
        # (one 64-deep contraction over the zero-padded batch), so the gradient is bit-equal to the one-product form.
        Bp = (B + 63) // 64 * 64
        st_p = torch.zeros(Bp, D, dtype=BF16, device=dev); st_p[:B] = ctx.emb.st
        mw_lo = (self.mod_w.data_ptr() - self.arena.data_ptr()) // 2
        mw_hi = mw_lo + self.mod_total * D
        mod_rows_lo = [self.mod_total]                    # rows [mod_rows_lo, mod_total) of dW_mod are written (and handed over)

        def mod_rows_grad(r0):
            r1 = mod_rows_lo[0]
            if r1 <= r0:
                return
            dp = torch.zeros(Bp, r1 - r0, dtype=BF16, device=dev); dp[:B] = dmod[:, r0:r1]
            ops.gemm_tn(dp, st_p, out=self.g_mod_w[r0:r1])
            mod_rows_lo[0] = r0
            if sync is not None and 0 <= mw_lo and mw_hi <= self._blocks_arena_lo:
                sync.ready(mw_lo + r0 * D, mw_lo + r1 * D)

        mod_rows_grad(self.mod_off_out)                   # norm_out's (scale, shift) rows: final since mod_grads above
        for li in range(len(self.blocks) - 1, -1, -1):
            if ctx.blocks[li] is None:
                self._recompute_segment(ctx, li)
            if li in ctx.route_end:                       # backward enters a TREAD route at its END: the routed blocks see only the kept tokens' gradient rows
                r_info = ctx.route_end[li]
                d_full = d_img
                d_img = ops.gather_rows(d_full.view(B, ctx.Si, D), r_info.keep_i32()).view(-1, D)
            Si, S, Sp = ctx.envs[li].Si, ctx.envs[li].S, ctx.envs[li].Sp
            blk, sv = self.blocks[li], ctx.blocks[li]
            ctx.blocks[li] = None
            if li + 1 < len(self.blocks):
                nb = self.blocks[li + 1]
                if sync is not None:
                    sync.ready(nb.arena_lo, nb.arena_hi)                      # the block processed last iteration: its slice can go out
                mod_rows_grad(nb.mod_off)                                     # ... and so can its rows of the modulation matrix
            mi = mod[:, blk.mod_off:blk.mod_off + 6 * D]; dmi = dmod[:, blk.mod_off:blk.mod_off + 6 * D]
            nct = 2 if blk.last else 6
            mt = mod[:, blk.mod_off_c:blk.mod_off_c + nct * D]; dmt = dmod[:, blk.mod_off_c:blk.mod_off_c + nct * D]
            if (_BLOCK_ABI and not blk.dual and ops.ATTN_TR and sv.Qt is None and d_img.is_contiguous() and (d_txt is None or d_txt.is_contiguous())
                    and blk.norm_q is None and blk.norm_k is None and blk.norm_added_q is None and blk.norm_added_k is None):
                # the data path as ONE C entry point (st355_block_sd3_joint_bwd), then every weight / bias / modulation gradient of the block from the gradients
                # it left behind (SD3.5's trainable q / k norm weights take the host-side form: their gradient rides in the RMSNorm backward pass)
                # (every bias / modulation-shift / -scale / gate gradient of the block is written by the entry itself — the column sums ride in its scale_cols and
                # LayerNorm-backward passes, csrc/stats.hip; only the weight gradients are left, taken from the gradients it kept)
                d_img, d_txt, G = self._block_bwd_c(blk, sv, ctx.envs[li], mod, cos, sin, d_img, d_txt, True, dmod=dmod)
                wgrad(blk.ff2, G.g_i, sv.h_i, bias=False)
                wgrad(blk.ff1, G.dh_i, sv.n2_i, bias=False)
                if not blk.last:
                    wgrad(blk.ffc2, G.g_t, sv.h_t, bias=False)
                    wgrad(blk.ffc1, G.dh_t, sv.n2_t, bias=False)
                wgrad(blk.to_out, G.dx1g_i, rows_of(sv.O, 0, Si, seg=True), bias=False)
                if not blk.last:
                    wgrad(blk.to_add_out, G.dx1g_t, rows_of(sv.O, Si, St), bias=False)
                wgrad(blk.qkv, G.c_img if G.c_img is not None else rows_of(G.dqkv, 0, Si, seg=True), sv.n_img, bias=False)       # (c_*: the compact copy the entry left)
                wgrad(blk.add_qkv, G.c_txt if G.c_txt is not None else rows_of(G.dqkv, Si, St), sv.n_txt, bias=False)
                del sv, G
                if li in ctx.route_start:
                    ops.scatter_rows(d_img.view(B, Si, D), ctx.route_start[li].keep_i32(), d_full.view(B, ctx.Si, D))
                    d_img, d_full = d_full, None
                continue
            # ---- MLP branch (host-sequenced form: SD3.5 dual-attention blocks, trainable q / k norms, TREAD, ST355_BLOCK_ABI=0).  Every modulation / gate / bias
            # gradient rides in the pass that streams its operands — the same fused entry points the C block sequences (csrc/stats.hip) ----
            g_i = ops.scale_cols_stats(d_img, mi[:, 5 * D:6 * D], Si, y_branch=sv.yf_i, d_gate=dmi[:, 5 * D:6 * D], d_bias=blk.ff2.gb)     # + d gate_mlp, d b_ff2
            wgrad(blk.ff2, g_i, sv.h_i, bias=False)
            dh_i = ops.gemm(g_i, blk.ff2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_img)
            wgrad(blk.ff1, dh_i, sv.n2_i, bias=False)
            dn2_i = ops.gemm(dh_i, blk.ff1.wT)
            dn2a = None
            if blk.dual:
                d2 = sv.d2
                mi9 = mod[:, blk.mod_off:blk.mod_off + 9 * D]; dmi9 = dmod[:, blk.mod_off:blk.mod_off + 9 * D]
                # + d shift_mlp, d scale_mlp, d gate_msa2 = sum d x1 * y_attn2, d b_to_out2 = sum gate_msa2 * d x1
                dx1_i, dxg2 = ops.ln_modulate_bwd_stats(dn2_i, sv.x1_img, mi[:, 4 * D:5 * D], Si, dmi[:, 3 * D:4 * D], dmi[:, 4 * D:5 * D], dres=d_img,
                                                        gate=mi9[:, 8 * D:9 * D], y_branch=d2.ya2, d_gate=dmi9[:, 8 * D:9 * D], d_bias=blk.to_out2.gb, want_gated=True)
                ops.colsum_rows(dh_i, Si, Si, B, blk.ff1.gb)
                wgrad(blk.to_out2, dxg2, d2.O2, bias=False)
                dO2 = ops.gemm(dxg2, blk.to_out2.wT)
                dqkv2 = torch.empty(B * Si, 3 * D, dtype=BF16, device=dev)
                dQ2 = torch.empty(B, H, Si, hd, dtype=BF16, device=dev); dK2 = torch.empty_like(dQ2)
                ops.attn_bwd(d2.Q2, d2.K2, d2.Q2t, d2.K2t, d2.qkv2[:, 2 * D:], d2.O2, dO2, d2.lse2b, dQ2, dK2, dqkv2[:, 2 * D:], B, H, Si, d2.Sip, hd, scale)
                qk_bwd(dQ2, dK2, d2.qkv2, blk.norm_q2, blk.norm_k2, dqkv2, Si, 0, Si, blk.g_norm_q2, blk.g_norm_k2)
                wgrad(blk.qkv2, dqkv2, d2.n2a)
                dn2a = ops.gemm(dqkv2, blk.qkv2.wT)
                dx1g_i = ops.scale_cols_stats(dx1_i, mi[:, 2 * D:3 * D], Si, y_branch=sv.ya_i, d_gate=dmi[:, 2 * D:3 * D], d_bias=blk.to_out.gb)   # + d gate_msa, d b_to_out
                del dxg2, dO2, dQ2, dK2, dqkv2, d2
            else:
                # + d shift_mlp, d scale_mlp, d gate_msa = sum d x1 * y_attn, d b_to_out = sum gate_msa * d x1
                dx1_i, dx1g_i = ops.ln_modulate_bwd_stats(dn2_i, sv.x1_img, mi[:, 4 * D:5 * D], Si, dmi[:, 3 * D:4 * D], dmi[:, 4 * D:5 * D], dres=d_img,
                                                          gate=mi[:, 2 * D:3 * D], y_branch=sv.ya_i, d_gate=dmi[:, 2 * D:3 * D], d_bias=blk.to_out.gb, want_gated=True)
                ops.colsum_rows(dh_i, Si, Si, B, blk.ff1.gb)
            del g_i, dh_i, dn2_i
            dx1_t = dx1g_t = None
            if not blk.last:
                g_t = ops.scale_cols_stats(d_txt, mt[:, 5 * D:6 * D], St, y_branch=sv.yf_t, d_gate=dmt[:, 5 * D:6 * D], d_bias=blk.ffc2.gb)
                wgrad(blk.ffc2, g_t, sv.h_t, bias=False)
                dh_t = ops.gemm(g_t, blk.ffc2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_txt)
                wgrad(blk.ffc1, dh_t, sv.n2_t, bias=False)
                dn2_t = ops.gemm(dh_t, blk.ffc1.wT)
                dx1_t, dx1g_t = ops.ln_modulate_bwd_stats(dn2_t, sv.x1_txt, mt[:, 4 * D:5 * D], St, dmt[:, 3 * D:4 * D], dmt[:, 4 * D:5 * D], dres=d_txt,
                                                          gate=mt[:, 2 * D:3 * D], y_branch=sv.ya_t, d_gate=dmt[:, 2 * D:3 * D], d_bias=blk.to_add_out.gb, want_gated=True)
                ops.colsum_rows(dh_t, St, St, B, blk.ffc1.gb)
                del g_t, dh_t, dn2_t
            # ---- attention output projections ----
            O_i = rows_of(sv.O, 0, Si)
            wgrad(blk.to_out, dx1g_i, O_i, bias=False)
            dO = (torch.zeros if blk.last else torch.empty)(B * S, D, dtype=BF16, device=dev)
            after = []
            probs = _stream_problems(B, S, Si, dict(a=dx1g_i, w=blk.to_out.wT, out=_rows3(dO, 0, Si, B, S)), after)
            if not blk.last:
                probs += _stream_problems(B, S, St, dict(a=dx1g_t, w=blk.to_add_out.wT, out=_rows3(dO, Si, St, B, S)), after)
            ops.gemm_grouped(probs)
            for f in after:
                f()
            if not blk.last:
                wgrad(blk.to_add_out, dx1g_t, rows_of(sv.O, Si, St), bias=False)
            del dx1g_i, dx1g_t, O_i
            # ---- attention ----
            dqkv = torch.empty(B * S, 3 * D, dtype=BF16, device=dev)
            dQ = torch.empty(B, H, S, hd, dtype=BF16, device=dev); dK = torch.empty_like(dQ)
            ops.attn_bwd(sv.Q, sv.K, sv.Qt, sv.Kt, sv.qkv[:, 2 * D:], sv.O, dO, sv.lse2, dQ, dK, dqkv[:, 2 * D:], B, H, S, Sp, hd, scale)
            qk_bwd(dQ, dK, sv.qkv, blk.norm_q, blk.norm_k, dqkv, Si, 0, S, blk.g_norm_q, blk.g_norm_k)
            qk_bwd(dQ, dK, sv.qkv, blk.norm_added_q, blk.norm_added_k, dqkv, St, Si, S, blk.g_norm_added_q, blk.g_norm_added_k)
            del dQ, dK, dO
            ops.colsum_rows(dqkv, Si, S, B, blk.qkv.gb)                    # d b_qkv / d b_add_qkv: each stream's rows of the joint dqkv, summed in place
            ops.colsum_rows(dqkv[Si:], St, S, B, blk.add_qkv.gb)
            dq_i, dq_t = rows_of(dqkv, 0, Si), rows_of(dqkv, Si, St)
            wgrad(blk.qkv, dq_i, sv.n_img, bias=False)
            wgrad(blk.add_qkv, dq_t, sv.n_txt, bias=False)
            dn_i, dn_t = ops.gemm_grouped([dict(a=dq_i, w=blk.qkv.wT), dict(a=dq_t, w=blk.add_qkv.wT)])
            d_img, _ = ops.ln_modulate_bwd_stats(dn_i, sv.img, mi[:, D:2 * D], Si, dmi[:, :D], dmi[:, D:2 * D], dres=dx1_i)      # + d shift_msa, d scale_msa
            if dn2a is not None:                  # the second reader of LN(img): attn2's modulated input (shift_msa2, scale_msa2)
                d_img, _ = ops.ln_modulate_bwd_stats(dn2a, sv.img, mod[:, blk.mod_off + 7 * D:blk.mod_off + 8 * D], Si, dmi9[:, 6 * D:7 * D], dmi9[:, 7 * D:8 * D],
                                                     dres=d_img)
            if blk.last:                          # AdaLayerNormContinuous: chunks (scale, shift)
                d_txt, _ = ops.ln_modulate_bwd_stats(dn_t, sv.txt, mt[:, :D], St, dmt[:, D:2 * D], dmt[:, :D], dres=None)
            else:
                d_txt, _ = ops.ln_modulate_bwd_stats(dn_t, sv.txt, mt[:, D:2 * D], St, dmt[:, :D], dmt[:, D:2 * D], dres=dx1_t)
            del dqkv, sv, dn_i, dn_t, dq_i, dq_t
            if li in ctx.route_start:
                ops.scatter_rows(d_img.view(B, Si, D), ctx.route_start[li].keep_i32(), d_full.view(B, ctx.Si, D))
                d_img, d_full = d_full, None
        Si, S, Sp = ctx.Si, ctx.S, ctx.Sp
        # ---- embedders ----
        em = ctx.emb
        wgrad(self.l_patch, d_img, em.patches)                              # PatchEmbed conv == GEMM on the patches; the position table is a buffer
        wgrad(self.l_ctx, d_txt, em.enc2d)
        # modulation linear: mod = silu(temb) W_mod^T + b
        dmod_p = torch.zeros(Bp, self.mod_total, dtype=BF16, device=dev); dmod_p[:B] = dmod
        mod_rows_grad(0)                                                    # block 0's rows (every other block's went out behind the block after it)
        tb = torch.empty(1, self.mod_total, dtype=F32, device=dev)
        ops.colsum_prod(dmod_p, tb)
        self.g_mod_b.copy_(tb[0])
        dmod_t = ops.transpose(dmod_p[:8 * ((B + 7) // 8)])                 # [mod_total, B8]
        dst = ops.transpose(ops.gemm_tn(self.mod_w, dmod_t))[:B].contiguous()   # d silu(temb) = dmod @ W_mod  ->  [B, D]
        dtemb = ops.silu_bwd(em.temb, dst)

        def mlp_bwd(l1, l2, x_in, pre1, act1, dy):
            """TimestepEmbedding / text projection: y = l2(silu(l1(x)))"""
            dyp = torch.zeros(Bp, dy.shape[1], dtype=BF16, device=dev); dyp[:B] = dy
            a1p = torch.zeros(Bp, act1.shape[1], dtype=BF16, device=dev); a1p[:B] = act1
            ops.gemm_tn(dyp, a1p, out=l2.gw)
            tb2 = torch.empty(1, dy.shape[1], dtype=F32, device=dev); ops.colsum_prod(dyp, tb2); l2.gb.copy_(tb2[0])
            d1 = ops.silu_bwd(pre1, ops.gemm(dy, l2.wT))
            d1p = torch.zeros(Bp, d1.shape[1], dtype=BF16, device=dev); d1p[:B] = d1
            xp = torch.zeros(Bp, x_in.shape[1], dtype=BF16, device=dev); xp[:B] = x_in
            ops.gemm_tn(d1p, xp, out=l1.gw)
            ops.colsum_prod(d1p, tb2); l1.gb.copy_(tb2[0])

        mlp_bwd(self.l_t1, self.l_t2, em.tproj, em.t1, em.st1, dtemb)
        mlp_bwd(self.l_p1, self.l_p2, em.pooled, em.p1, em.sp1, dtemb)
        if sync is not None:
            if 0 <= mw_lo and mw_hi <= self._blocks_arena_lo:                 # embedders, modulation bias, block 0: what lies around the modulation matrix's rows
                sync.ready(mw_hi, self.blocks[0].arena_hi)
                sync.ready(0, mw_lo)
            else:
                sync.ready(0, self.blocks[0].arena_hi)
        return None

    # ------------------------------------------------------------------------------------------------
    # public forward (reference signature: sd3/transformer.py:560-575)
    # ------------------------------------------------------------------------------------------------
    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, block_controlnet_hidden_states=None,
                joint_attention_kwargs=None, return_dict: bool = True, force_keep_mask=None, **unsupported):
        self._force_keep_mask = force_keep_mask            # TREAD: tokens that may never be routed away (sd3/transformer.py:571, 699-703)
        if block_controlnet_hidden_states is not None:
            raise NotImplementedError("SD3 ControlNet residuals are not wired to the st355 path yet")
        for k, v in unsupported.items():
            if v is not None and v is not False:
                raise NotImplementedError(f"SD3Transformer2DModel(st355): argument {k!r} is not supported on the HIP path")
        if timestep.ndim not in (1, 2):
            raise ValueError(f"timestep: expected [B] or tokenwise [B, S_img], got {tuple(timestep.shape)}")
        need_grad = torch.is_grad_enabled() and (len(self._lora_params) > 0 or getattr(self, "full", False))
        if need_grad and not self._prepared and not getattr(self, "full", False):
            self.prepare_for_training()      # K-major dgrad operands went stale (new weights / replica start-state broadcast): rebuild lazily
        if need_grad and getattr(self, "full", False):
            out = _SD3FullFn.apply(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, *self._full_params)
        elif need_grad:
            out = _SD3Fn.apply(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, *self._lora_params)
        else:
            with torch.no_grad():
                out, _ = self._engine_forward(hidden_states.to(BF16), encoder_hidden_states.to(BF16), pooled_projections, timestep, save=False)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)


class _SD3Fn(torch.autograd.Function):
    """one autograd node for the whole network (see flux/transformer.py::_FluxFn)"""

    @staticmethod
    def forward(fctx, model, latents, enc, pooled, timestep, *lora_params):
        out, ctx = model._engine_forward(latents.detach().to(BF16), enc.detach().to(BF16), pooled.detach(), timestep.detach(), save=True)
        fctx.model, fctx.ectx = model, ctx
        return out

    @staticmethod
    def backward(fctx, dout):
        model = fctx.model
        if model.grad_sync is not None:
            model.grad_sync.begin()
        model._engine_backward(fctx.ectx, dout)
        fctx.ectx = None
        if model.grad_sync is not None:
            model.grad_scale_from_sync = model.grad_sync.finish()
        from ..training.grad_sync import hand_over_gradients
        gflat = hand_over_gradients(model, model.lora_grad_flat)
        model._last_grad_flat = gflat
        grads, off = [], 0
        for p in model._lora_params:
            n = p.numel()
            grads.append(gflat[off:off + n].view_as(p))
            off += n
        return (None,) * 5 + tuple(grads)


class _SD3FullFn(torch.autograd.Function):
    """full fine-tune: one autograd node; backward fills the bf16 gradient arena and hands autograd views of a private copy"""

    @staticmethod
    def forward(fctx, model, latents, enc, pooled, timestep, *params):
        out, ctx = model._engine_forward(latents.detach().to(BF16), enc.detach().to(BF16), pooled.detach(), timestep.detach(), save=True, full=True)
        fctx.model, fctx.ectx = model, ctx
        return out

    @staticmethod
    def backward(fctx, dout):
        model = fctx.model
        model._pick_grad_arena()
        if model.grad_sync is not None:
            model.grad_sync.begin()
        model._engine_backward_full(fctx.ectx, dout)
        fctx.ectx = None
        if model.grad_sync is not None:
            model.grad_scale_from_sync = model.grad_sync.finish()   # every slice reduced (SUM over replicas); the optimizer folds 1/world
        from ..training.grad_sync import hand_over_gradients
        gflat = hand_over_gradients(model, model.grad_arena)
        model._last_grad_flat = gflat
        grads = [gflat[off:off + n].view_as(p) for (off, n), p in zip(model._full_offsets, model._full_params)]
        return (None,) * 5 + tuple(grads)
