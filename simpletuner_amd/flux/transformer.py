"""FluxTransformer2DModel — the MI355X-native trained component for the Flux.1 MMDiT.

Mirrors the reference's module surface (simpletuner/helpers/models/flux/transformer.py:690-1513): same constructor
arguments, same `forward(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
guidance, ..., return_dict)` contract, same state-dict keys as the diffusers checkpoint, `.config`, `.parameters()`,
DDP-wrappable, usable as `ModelFoundation.MODEL_CLASS`.  Nothing here is a torch op on the hot path: forward and the
hand-written backward are sequences of libst355 launches (simpletuner_amd.ops); torch owns memory, streams and autograd's
outer edge (one autograd.Function for the whole network, so `accelerator.backward(loss)` works unchanged).

Storage layout (designed for 288 GB HBM, not for a 24 GB card):
  * projection weights that share an input live FUSED in one buffer ([to_q|to_k|to_v] -> [3D,D]); the per-projection
    nn.Parameters are views into it, so checkpoints load straight into the fused layout and nothing is duplicated;
  * all AdaLN modulation linears of all 57 blocks (+norm_out) are one [342D+2D, D] matrix: the per-step modulation
    is ONE skinny GEMM on SiLU(temb) instead of 115 launches;
  * frozen-base (LoRA) training keeps a K-major transposed copy of every weight used by a dgrad GEMM (W^T), so the
    backward reuses the same NT kernel (costs +1x weights of HBM; 2 x 24 GB for Flux.1-dev);
  * LoRA adapters live in two flat fp32 arenas (params, grads) -> one fused AdamW(+EMA) launch and one RCCL all-reduce;
  * every base parameter is a view of ONE bf16 arena; full-rank training (`enable_full_finetune`, `_engine_backward_full`) adds a gradient arena of the
    same layout: one fused optimizer launch over 12 B parameters, contiguous per-block slices for the gradient exchange.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..ops import EPI_ADD, EPI_GATE_RESIDUAL, EPI_GELU, EPI_MUL_GELU_GRAD, EPI_NONE, EPI_QK_NORM_ROPE

BF16 = torch.bfloat16
F32 = torch.float32
_FUSED_QKV = __import__("os").environ.get("ST355_FUSED_QKV", "1") != "0"      # A/B switch: 0 = separate RMSNorm + RoPE pass after the QKV projection
_FUSED_ROPE_BWD = __import__("os").environ.get("ST355_FUSED_ROPE_BWD", "1") != "0"   # A/B switch: 0 = RoPE / RMSNorm backward as its own pass after the attention backward
_FUSED_VT = __import__("os").environ.get("ST355_FUSED_VT", "1") != "0"        # A/B switch: 0 = no V^T from the fused epilogue, forward attention reads row-major V
_BLOCK_ABI = __import__("os").environ.get("ST355_BLOCK_ABI", "1") != "0"          # A/B switch: 0 = sequence the blocks' kernels from the host instead of st355_block_flux_*
_BLOCK_ABI_ONLY = __import__("os").environ.get("ST355_BLOCK_ABI_ONLY", "")           # debugging aid: "single" / "double" / "fwd" / "bwd" restricts the C entry points to that subset


def _block_abi_ok() -> bool:
    return _BLOCK_ABI
_TRANSPOSED_COPIES = __import__("os").environ.get("ST355_ATTN_BWD_T") == "1"      # A/B switch: keep the pre-transposed Q^T / K^T copies (dkv2 / dq kernels) at head_dim 128


# ------------------------------------------------------------------------------------------------
# helpers to register parameters under dotted (checkpoint) names
# ------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    pass


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Holder())
        mod = getattr(mod, p)
    mod.register_parameter(parts[-1], param)


def _frozen(t: torch.Tensor) -> nn.Parameter:
    return nn.Parameter(t, requires_grad=False)


class LoraGroup:
    """LoRA adapters of the projections that share one input (one fused GEMM).  peft semantics: y += (alpha/r) B A x."""

    def __init__(self, K: int, N_total: int, targets: List[Tuple[str, int, int]], rank: int, alpha: float, device):
        self.K, self.N_total, self.targets = K, N_total, targets
        self.rank, self.scale = rank, alpha / rank
        # adapter columns inside the K-extension: 32 / 64 (one pass of the rank-space kernels), above 64 a multiple of 64 walked in 64-column slabs
        # (the reference's sd3.peft-lora example trains rank 128)
        self.r_pad = 32 if rank <= 32 else (rank + 63) // 64 * 64
        self.K2 = (len(targets) * self.r_pad + 63) // 64 * 64
        self.k2_real = len(targets) * rank          # adapter columns inside the padded extension (algorithmic-work accounting of the profiler)
        z = lambda *s: torch.zeros(*s, dtype=BF16, device=device)
        self.A_cat, self.A_cat_T = z(self.K2, K), z(K, self.K2)
        self.B_blk, self.B_blk_T = z(N_total, self.K2), z(self.K2, N_total)
        self.A: List[torch.Tensor] = []   # fp32 params (views into the flat arena), filled by the owner
        self.B: List[torch.Tensor] = []
        self.gA: List[torch.Tensor] = []  # fp32 grad views
        self.gB: List[torch.Tensor] = []
        self.flat_lo = self.flat_hi = 0   # this group's [lo, hi) element range inside the flat gradient arena

    def pack(self):
        for g, (_, n_off, _) in enumerate(self.targets):
            ops.lora_pack(self.A[g], self.B[g], self.scale, self.A_cat, self.A_cat_T, self.B_blk, self.B_blk_T,
                          k2_off=g * self.r_pad, n_off=n_off)

    def grads(self, x, T, dy, U, accumulate: bool, sync=None):
        """dB_g = s * dy_g^T T_g ; dA_g = U_g^T x   (rank-space backward: both products are [*, r]).  x: the projection's input, or — an input that only
        exists as K segments (the single block's proj_out reads [attn | mlp] as a two-segment K loop) — a list of (segment, first input column)."""
        segs = x if isinstance(x, (list, tuple)) else [(x, 0)]
        multi = len(segs) == 1 and 1 < len(self.targets) <= 4 and self.r_pad == 32 and U.shape[1] >= 128   # q / k / v share x: dA of all three in ONE pass over x
        cw = min(self.r_pad, 64)                                 # rank-space kernels take 32 or 64 adapter columns per pass
        for g, (_, n_off, N) in enumerate(self.targets):
            for s0 in range(0, self.rank, cw):
                c0, r_used = g * self.r_pad + s0, min(cw, self.rank - s0)
                ops.skinny_tn(dy[..., n_off:n_off + N], T[:, c0:c0 + cw], self.gB[g][:, s0:], self.rank, 1, r_used, alpha=self.scale, accumulate=accumulate)
                if not multi:
                    for (xs, k0) in segs:
                        ops.skinny_tn(xs, U[:, c0:c0 + cw], self.gA[g][s0:, k0:], 1, self.K, r_used, alpha=1.0, accumulate=accumulate)
        if multi:
            ops.skinny_tn_multi(x, U, self.gA, 1, self.K, self.rank, alpha=1.0, accumulate=accumulate)
        if sync is not None:
            sync.ready(self.flat_lo, self.flat_hi)


class FluxTransformer2DModel(nn.Module):
    def __init__(self, patch_size: int = 1, in_channels: int = 64, num_layers: int = 19, num_single_layers: int = 38,
                 attention_head_dim: int = 128, num_attention_heads: int = 24, joint_attention_dim: int = 4096,
                 pooled_projection_dim: int = 768, guidance_embeds: bool = False, axes_dims_rope: Tuple[int, ...] = (16, 56, 56),
                 device=None, **_ignored):
        super().__init__()
        self.config = SimpleNamespace(patch_size=patch_size, in_channels=in_channels, num_layers=num_layers,
                                      num_single_layers=num_single_layers, attention_head_dim=attention_head_dim,
                                      num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
                                      pooled_projection_dim=pooled_projection_dim, guidance_embeds=guidance_embeds,
                                      axes_dims_rope=tuple(axes_dims_rope))
        if attention_head_dim not in (64, 128):
            raise ValueError("attention_head_dim must be 64 or 128 (kernels built for these)")
        self.out_channels = in_channels
        self.H, self.hd = num_attention_heads, attention_head_dim
        self.D = D = self.H * self.hd
        self.inner_dim = D
        if D % 64 or joint_attention_dim % 64 or pooled_projection_dim % 64 or in_channels % 64:
            raise ValueError("all contraction dims must be multiples of 64")
        self.device_ = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        dev = self.device_
        # every weight / bias / norm weight / modulation row is a view of ONE bf16 arena (allocation order = arena order): full-rank training then has one
        # gradient arena of the same layout, ONE fused optimizer launch and contiguous slices for the gradient exchange (as in sd3/transformer.py).  Pass 1
        # counts on meta tensors, pass 2 hands out the views.  Every tensor of Flux.1 holds a multiple of 64 elements, so all views start 128-byte aligned.
        self._arena_numel, self._counting = 0, True
        self._build()
        self.arena = torch.zeros(self._arena_numel, dtype=BF16, device=dev)
        self._arena_numel, self._counting = 0, False
        self._build()

        self.lora_groups: List[LoraGroup] = []
        self.mod_lora: Optional[LoraGroup] = None          # 'ai-toolkit': adapters on every block's AdaLN modulation Linear (one group over the fused modulation GEMM)
        self._dmod: Optional[torch.Tensor] = None          # d loss / d (modulation rows) of the backward in flight (only with mod_lora)
        self.lora_flat: Optional[torch.Tensor] = None
        self.lora_grad_flat: Optional[torch.Tensor] = None
        self._lora_params: List[nn.Parameter] = []
        self._rope_cache: Dict = {}
        self._rope_layout_key = None
        self._norm_w_ok: Dict = {}
        self._prepared = False
        self.accumulate_lora_grads = False
        self.gradient_checkpointing = False
        self.gradient_checkpointing_interval = None          # flux/transformer.py:816-818
        self.gradient_checkpointing_segment_stride = None
        self.gradient_checkpointing_backend = "torch"
        self._tread_router, self._tread_routes, self._force_keep_mask = None, None, None
        self.grad_sync = None            # training.grad_sync.GradSync over lora_grad_flat / grad_arena (data-parallel replicas)
        self._last_grad_flat = None
        self.full = False                # enable_full_finetune(): every base parameter trains

    def _alloc(self, *shape):
        n = 1
        for d in shape:
            n *= d
        off = self._arena_numel
        self._arena_numel += (n + 7) // 8 * 8               # every tensor starts 16-byte aligned inside the arena
        if self._counting:
            return torch.empty(*shape, dtype=BF16, device="meta")
        return self.arena[off:off + n].view(*shape)

    def _reg(self, name, param):
        if not self._counting:
            _attach(self, name, param)

    def _build(self):
        c = self.config
        D, dev, e = self.D, self.device_, self._alloc
        patch_size, in_channels, num_layers, num_single_layers = c.patch_size, c.in_channels, c.num_layers, c.num_single_layers
        joint_attention_dim, pooled_projection_dim, guidance_embeds = c.joint_attention_dim, c.pooled_projection_dim, c.guidance_embeds

        # ---- embedders ----
        def lin(name, out_f, in_f):
            w, b = e(out_f, in_f), e(out_f)
            self._reg(name + ".weight", _frozen(w)); self._reg(name + ".bias", _frozen(b))
            return SimpleNamespace(w=w, b=b, wT=None, lora=None)

        self.l_x = lin("x_embedder", D, in_channels)
        self.l_ctx = lin("context_embedder", D, joint_attention_dim)
        self.l_t1 = lin("time_text_embed.timestep_embedder.linear_1", D, 256)
        self.l_t2 = lin("time_text_embed.timestep_embedder.linear_2", D, D)
        if guidance_embeds:
            self.l_g1 = lin("time_text_embed.guidance_embedder.linear_1", D, 256)
            self.l_g2 = lin("time_text_embed.guidance_embedder.linear_2", D, D)
        self.l_p1 = lin("time_text_embed.text_embedder.linear_1", D, pooled_projection_dim)
        self.l_p2 = lin("time_text_embed.text_embedder.linear_2", D, D)

        # ---- one matrix for every AdaLN modulation linear ----
        self.mod_total = (num_layers * 12 + num_single_layers * 3 + 2) * D
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
            w = e(self.hd)
            if not self._counting:
                w.fill_(1.0)
            self._reg(name + ".weight", _frozen(w))
            return w

        self.double: List[SimpleNamespace] = []
        for i in range(num_layers):
            p = f"transformer_blocks.{i}."
            blk = SimpleNamespace(arena_lo=self._arena_numel)
            blk.mod_off = mod_slice(p + "norm1.linear", 6 * D)
            blk.mod_off_c = mod_slice(p + "norm1_context.linear", 6 * D)
            blk.qkv = fused(p + "attn.", ["to_q", "to_k", "to_v"], D, D)
            blk.add_qkv = fused(p + "attn.", ["add_q_proj", "add_k_proj", "add_v_proj"], D, D)
            blk.to_out = fused(p + "attn.", ["to_out.0"], D, D)
            blk.to_add_out = fused(p + "attn.", ["to_add_out"], D, D)
            blk.norm_q, blk.norm_k = normw(p + "attn.norm_q"), normw(p + "attn.norm_k")
            blk.norm_added_q, blk.norm_added_k = normw(p + "attn.norm_added_q"), normw(p + "attn.norm_added_k")
            blk.ff1 = fused(p, ["ff.net.0.proj"], 4 * D, D)
            blk.ff2 = fused(p, ["ff.net.2"], D, 4 * D)
            blk.ffc1 = fused(p, ["ff_context.net.0.proj"], 4 * D, D)
            blk.ffc2 = fused(p, ["ff_context.net.2"], D, 4 * D)
            blk.arena_hi = self._arena_numel
            self.double.append(blk)
        self.single: List[SimpleNamespace] = []
        for i in range(num_single_layers):
            p = f"single_transformer_blocks.{i}."
            blk = SimpleNamespace(arena_lo=self._arena_numel)
            blk.mod_off = mod_slice(p + "norm.linear", 3 * D)
            blk.qkv = fused(p + "attn.", ["to_q", "to_k", "to_v"], D, D)
            blk.norm_q, blk.norm_k = normw(p + "attn.norm_q"), normw(p + "attn.norm_k")
            blk.proj_mlp = fused(p, ["proj_mlp"], 4 * D, D)
            blk.proj_out = fused(p, ["proj_out"], D, 5 * D)
            blk.arena_hi = self._arena_numel
            self.single.append(blk)
        self.mod_off_out = mod_slice("norm_out.linear", 2 * D)
        assert off == self.mod_total
        self._head_arena_lo = self._arena_numel
        self.l_out = lin("proj_out", patch_size * patch_size * self.out_channels, D)

    # ------------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def load_flat_state(self, state: Dict[str, torch.Tensor]):
        """copy a {checkpoint name: tensor} dict into the fused buffers (names = diffusers state-dict keys)"""
        own = dict(self.named_parameters())
        missing = [k for k in own if k not in state and ".lora_" not in k]
        if missing:
            raise KeyError(f"missing weights: {missing[:5]} ... ({len(missing)})")
        for k, v in state.items():
            if k in own:
                own[k].data.copy_(v.to(device=own[k].device, dtype=own[k].dtype))
        self._prepared = False
        self._norm_w_ok.clear()

    @torch.no_grad()
    def init_synthetic(self, seed: int = 42):
        """seed-deterministic random init on device, same distribution family as oracle.flux.init_params (benchmarks)."""
        g = torch.Generator(device=self.device_).manual_seed(seed)
        for name, p in self.named_parameters():
            if ".lora_" in name:
                continue
            if "norm_q" in name or "norm_k" in name or "norm_added" in name:
                p.data.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=self.device_))
            elif name.endswith(".bias"):
                p.data.copy_(0.02 * torch.randn(p.shape, generator=g, device=self.device_))
            else:
                p.data.copy_(torch.randn(p.shape, generator=g, device=self.device_, dtype=BF16) * (1.0 / math.sqrt(p.shape[1])))
        self._prepared = False
        self._norm_w_ok.clear()

    @torch.no_grad()
    def prepare_for_training(self):
        """build the K-major transposed copies the dgrad GEMMs read (frozen base => done once)."""
        def tr(l):
            l.wT = l.w.t().contiguous()
        for blk in self.double:
            for l in (blk.qkv, blk.add_qkv, blk.to_out, blk.to_add_out, blk.ff1, blk.ff2, blk.ffc1, blk.ffc2):
                tr(l)
        for blk in self.single:
            for l in (blk.qkv, blk.proj_mlp, blk.proj_out):
                tr(l)
        tr(self.l_out)
        if getattr(self, "full", False):            # full-rank training also back-propagates through the conditioning MLPs' second linears
            for l in (self.l_t2, self.l_p2) + ((self.l_g2,) if self.config.guidance_embeds else ()):
                tr(l)
        self._prepared = True

    # ------------------------------------------------------------------------------------------------
    # LoRA (peft-compatible naming: <module>.lora_A.default.weight / lora_B.default.weight)
    # ------------------------------------------------------------------------------------------------
    def add_lora_adapter(self, rank: int = 32, alpha: Optional[float] = None, targets: str = "default", seed: int = 7,
                         init_b_std: float = 0.0):
        """common.py:1049-1128 (LoraConfig(r, lora_alpha, target_modules)).  targets: 'default' = attn to_q,to_k,to_v,to_out.0
        (+ single-block to_q,to_k,to_v); 'all' adds the context-stream projections; 'context' = only those (flux/model.py:1263-1271); 'all+ffs' / 'context+ffs' add
        the feed-forward Linears of that set's streams (flux/model.py:1272-1301: ff.net.*, ff_context.net.*, the single blocks' proj_mlp / proj_out); 'tiny' / 'nano' =
        single_transformer_blocks.{7, 20}.proj_out / .7.proj_out only (flux/model.py:1363-1375).  A block that carries a feed-forward adapter is sequenced from the
        host (the block-level C entry points know the attention adapters only)."""
        alpha = float(rank if alpha is None else alpha)
        D, dev = self.D, self.device_
        plan = []  # (group, name, n_off, N, K)
        self.lora_groups = []

        def group(lin, prefix, names, K):
            g = LoraGroup(K, lin.w.shape[0], [(prefix + n, j * (lin.w.shape[0] // len(names)), lin.w.shape[0] // len(names))
                                              for j, n in enumerate(names)], rank, alpha, dev)
            lin.lora = g
            self.lora_groups.append(g)
            for (name, n_off, N) in g.targets:
                plan.append((g, name, N, K))

        sets = ("default", "all", "context", "all+ffs", "context+ffs", "all+ffs+embedder", "ai-toolkit", "tiny", "nano")
        if targets not in sets:
            raise ValueError(f"add_lora_adapter: unknown target set {targets!r} (built: {', '.join(repr(t) for t in sets)})")
        if targets == "all+ffs+embedder":          # flux/model.py:1320-1339: all+ffs + x_embedder (the packed-latent input projection, K = in_channels)
            group(self.l_x, "", ["x_embedder"], self.l_x.w.shape[1])
            targets = "all+ffs"
        self.mod_lora = None
        if targets == "ai-toolkit":                # flux/model.py:1340-1362: all+ffs + the AdaLN modulation Linears norm1.linear / norm1_context.linear / norm.linear
            # every block's modulation is a row slice of ONE fused GEMM over silu(temb): its adapters are one group that shares that input (like q / k / v share theirs).
            # The K-extension operand is block-diagonal and stored dense: [mod_total, 76 * 32] bf16 = 5.2 GB for Flux.1 (+ its transpose) — HBM holds it; the GEMM has
            # M = batch rows, so the extra read is ~1 ms per step
            slices = []
            for i, blk in enumerate(self.double):
                slices += [(f"transformer_blocks.{i}.norm1.linear", blk.mod_off, 6 * D), (f"transformer_blocks.{i}.norm1_context.linear", blk.mod_off_c, 6 * D)]
            for i, blk in enumerate(self.single):
                slices.append((f"single_transformer_blocks.{i}.norm.linear", blk.mod_off, 3 * D))
            g = LoraGroup(D, self.mod_total, slices, rank, alpha, dev)
            self.mod_lora = g
            self.lora_groups.append(g)
            for (name, n_off, N) in g.targets:
                plan.append((g, name, N, D))
            targets = "all+ffs"
        base, ffs = (targets[:-4], True) if targets.endswith("+ffs") else (targets, False)
        if base == "context" and not self.double:
            raise ValueError(f"add_lora_adapter: the {targets!r} target set names the double blocks' context-stream layers; this model has no double block")
        if targets in ("tiny", "nano"):            # flux/model.py:1363-1375: single_transformer_blocks.7(.20).proj_out — nothing else carries an adapter
            want = (7, 20) if targets == "tiny" else (7,)
            if max(want) >= len(self.single):
                raise ValueError(f"add_lora_adapter: the {targets!r} target set names single_transformer_blocks.{max(want)}.proj_out; this model has {len(self.single)} single blocks")
            for i in want:
                group(self.single[i].proj_out, f"single_transformer_blocks.{i}.", ["proj_out"], 5 * D)
        else:
            for i, blk in enumerate(self.double):
                b = f"transformer_blocks.{i}."
                p = b + "attn."
                if base != "context":
                    group(blk.qkv, p, ["to_q", "to_k", "to_v"], D)
                    group(blk.to_out, p, ["to_out.0"], D)
                if base in ("all", "context"):       # flux/model.py:1263-1271 "context": add_q/k/v_proj + to_add_out only (the single blocks have no such layers)
                    group(blk.add_qkv, p, ["add_q_proj", "add_k_proj", "add_v_proj"], D)
                    group(blk.to_add_out, p, ["to_add_out"], D)
                if ffs and base == "all":             # flux/model.py:1283-1301 "all+ffs": + ff.net.0.proj, ff.net.2, ff_context.*, proj_mlp, proj_out
                    group(blk.ff1, b, ["ff.net.0.proj"], D)
                    group(blk.ff2, b, ["ff.net.2"], 4 * D)
                if ffs:                               # flux/model.py:1272-1282 "context+ffs": + ff_context.net.0.proj, ff_context.net.2
                    group(blk.ffc1, b, ["ff_context.net.0.proj"], D)
                    group(blk.ffc2, b, ["ff_context.net.2"], 4 * D)
            if base != "context":
                for i, blk in enumerate(self.single):
                    b = f"single_transformer_blocks.{i}."
                    group(blk.qkv, b + "attn.", ["to_q", "to_k", "to_v"], D)
                    if ffs:
                        group(blk.proj_mlp, b, ["proj_mlp"], D)
                        group(blk.proj_out, b, ["proj_out"], 5 * D)
                if ffs:                               # peft wraps every module whose name ends with an entry: "proj_out" is also the model's output projection
                    group(self.l_out, "", ["proj_out"], D)
        # the backward stops below the first block (execution order: double stack, then single stack) that carries an adapter: nothing upstream of it trains
        # ('tiny' / 'nano': single block 7 — the 19 double and 7 single blocks below it run no backward at all)
        carries = [any(l.lora is not None for l in (b.qkv, b.add_qkv, b.to_out, b.to_add_out, b.ff1, b.ff2, b.ffc1, b.ffc2)) for b in self.double] \
            + [any(l.lora is not None for l in (b.qkv, b.proj_mlp, b.proj_out)) for b in self.single]
        self._bwd_stop = carries.index(True) if True in carries else 0
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
            g.flat_hi = off + rank * K + N * rank
            a = self.lora_flat[off:off + rank * K].view(rank, K); ga = self.lora_grad_flat[off:off + rank * K].view(rank, K)
            off += rank * K
            b = self.lora_flat[off:off + N * rank].view(N, rank); gb = self.lora_grad_flat[off:off + N * rank].view(N, rank)
            off += N * rank
            bound = 1.0 / math.sqrt(K)   # kaiming_uniform(a=sqrt(5)) on [r,K] (peft default for lora_A)
            a.copy_((torch.rand(rank, K, generator=gen, device=dev) * 2 - 1) * bound)
            if init_b_std > 0:
                b.copy_(torch.randn(N, rank, generator=gen, device=dev) * init_b_std)
            pa, pb = nn.Parameter(a), nn.Parameter(b)
            _attach(self, name + ".lora_A.default.weight", pa); _attach(self, name + ".lora_B.default.weight", pb)
            g.A.append(pa.data); g.B.append(pb.data); g.gA.append(ga); g.gB.append(gb)
            self._lora_params += [pa, pb]
        return self._lora_params

    def trainable_parameters(self):
        return list(self._full_params) if self.full else list(self._lora_params)

    # ------------------------------------------------------------------------------------------------
    # rope tables (FluxPosEmbed, theta=1e4, float64 frequencies -> fp32 tables); cached per id layout
    # ------------------------------------------------------------------------------------------------
    _ROPE_CACHE_MAX = 32

    def _rope(self, txt_ids: torch.Tensor, img_ids: torch.Tensor):
        """Tables for the id layout of this call.  Cached only under an explicit layout key the caller supplies (`forward(rope_layout_key=...)`: the plugin
        names the layout by what it was built from — latent grid, text length, a digest of the reference-image ids).  Never keyed on tensor identity: the
        caching allocator hands a freshly concatenated id tensor the same address step after step while its contents change (transposed aspect buckets,
        reference images of another size with the same token count).  Without a key the tables are computed on the device from the ids themselves: no
        device->host read (a host sync per step, and illegal while a hipGraph is being captured), nothing retained."""
        key = self._rope_layout_key
        if key is not None:
            key = (key, tuple(txt_ids.shape), tuple(img_ids.shape))
            hit = self._rope_cache.get(key)
            if hit is not None:
                return hit
        ids = torch.cat((txt_ids.to(self.device_), img_ids.to(self.device_)), dim=0).to(torch.float64)
        cos_l, sin_l = [], []
        for i, d in enumerate(self.config.axes_dims_rope):
            freqs = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float64, device=ids.device)[: d // 2] / d))
            f = torch.outer(ids[:, i], freqs)
            cos_l.append(f.cos().repeat_interleave(2, dim=1).float()); sin_l.append(f.sin().repeat_interleave(2, dim=1).float())
        cos, sin = torch.cat(cos_l, -1).contiguous(), torch.cat(sin_l, -1).contiguous()
        # + one (cos, sin) per interleaved pair, [S, hd/2]: what the fused projection epilogue reads (half the table bytes through the CU's L2 port)
        out = (cos, sin, cos[:, 0::2].contiguous(), sin[:, 0::2].contiguous())
        if key is not None:
            while len(self._rope_cache) >= self._ROPE_CACHE_MAX:          # bounded: oldest layout out
                self._rope_cache.pop(next(iter(self._rope_cache)))
            self._rope_cache[key] = out
        return out

    # ------------------------------------------------------------------------------------------------
    # forward / backward engines
    # ------------------------------------------------------------------------------------------------
    def _lin_fwd(self, lin, x, **kw):
        """y = x W^T + b (+ LoRA K-extension).  Returns (y, T) where T = x A^T (kept for the rank-space backward)."""
        T = None
        if lin.lora is not None:
            T = ops.gemm(x, lin.lora.A_cat)
            kw.update(a2=T, b2=lin.lora.B_blk, k2_real=lin.lora.k2_real)
        return ops.gemm(x, lin.w, bias=lin.b, **kw), T

    def _lin_bwd(self, lin, dy, x=None, T=None, **kw):
        """dx = dy W (+ LoRA K-extension (dy sB) A); also writes the adapter gradients when the projection has one."""
        U = None
        if lin.lora is not None:
            U = ops.gemm(dy, lin.lora.B_blk_T)
            kw.update(a2=U, b2=lin.lora.A_cat_T, k2_real=lin.lora.k2_real)
        dx = ops.gemm(dy, lin.wT, **kw)
        if lin.lora is not None:
            lin.lora.grads(x, T, dy, U, self.accumulate_lora_grads, self.grad_sync)
        return dx

    # ------------------------------------------------------------------------------------------------
    # activation checkpointing (flux/transformer.py:816-835, 1142-1209, gradient_checkpointing_interval.py:51-120)
    # ------------------------------------------------------------------------------------------------
    def set_gradient_checkpointing_interval(self, value):
        self.gradient_checkpointing_interval = None if value is None else int(value)

    def set_gradient_checkpointing_segment_stride(self, segment_stride):
        self.gradient_checkpointing_segment_stride = None if segment_stride is None else int(segment_stride)

    def set_gradient_checkpointing_backend(self, backend: str):
        if backend != "torch":      # "unsloth" = CPU-offloaded checkpoints (pointless with 288 GB), "*-ffn" = FFN-only scope: not built, never silently ignored
            raise NotImplementedError(f"gradient_checkpointing_backend={backend!r} is not implemented on the st355 path (built: 'torch' = recompute)")
        self.gradient_checkpointing_backend = backend

    def enable_gradient_checkpointing(self, *a, **k):
        self.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    def _checkpoint_segments(self, n_blocks: int):
        """[(first block, block count, recompute?)] over one block stack, the reference's three modes:
             gradient_checkpointing off                  -> every block keeps its activations;
             on, interval None / <= 1  ("layer")         -> every block is its own checkpoint (only its input is kept, the block is re-run in backward);
             on, interval k > 1 [, segment_stride s >= k] -> the first k blocks of every s-block window form ONE checkpoint (only the segment input is
                                                            kept), the s - k blocks of the gap keep their activations  (`checkpoint_sequential_state`)."""
        from ..training.checkpoint_plan import segments
        return segments(n_blocks, self.gradient_checkpointing, self.gradient_checkpointing_interval, self.gradient_checkpointing_segment_stride)

    # ------------------------------------------------------------------------------------------------
    # forward / backward engines
    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _rows_of(joint, lo: int, rows: int, env):
        """rows [lo, lo + rows) of every sample of a joint [B * S, C] buffer, as a GEMM / skinny operand: a [B, rows, C] strided view (no copy)"""
        return joint.view(env.B, env.S, -1)[:, lo:lo + rows]

    @staticmethod
    def _problems(env, rows: int, pr: dict):
        """One projection over the `rows`-row block of every sample.  Operands may be compact [B * rows, C] tensors or [B, rows, C] views of joint
        buffers (_rows_of).  When the blocks are tile-aligned (rows % 256 == 0) that is ONE segmented problem (st355_gemm_args.seg_rows: one grid of
        B * rows / 256 row tiles instead of B launches that each fill the 256 CUs badly); otherwise one problem per sample, as before."""
        B = env.B
        ROWED = ("a", "a2", "out", "aux_in", "aux_out")
        if B == 1:
            return [{k: (v[0] if k in ROWED and torch.is_tensor(v) and v.dim() == 3 else v) for k, v in pr.items()}]
        if rows % 256 == 0:
            return [pr]
        out = []
        for b in range(B):
            q = {}
            for k, v in pr.items():
                if k in ROWED and torch.is_tensor(v):
                    q[k] = v[b] if v.dim() == 3 else v[b * rows:(b + 1) * rows]
                elif k == "gate":          # one gate row per sample, or — tokenwise timesteps — one per token row of this stream
                    q[k] = v[b:b + 1] if v.shape[0] == B else v[b * rows:(b + 1) * rows]
                else:
                    q[k] = v
            out.append(q)
        return out

    @staticmethod
    def _compact(t, env, rows: int):
        """a [B, rows, C] view as an operand of the rank-space gradient kernels: as is when they can walk it segmented, else a compact copy"""
        if t.dim() != 3:
            return t
        if env.B == 1:
            return t[0]
        return t if rows % 256 == 0 else t.reshape(env.B * rows, -1)

    def _fused_qkv_ok(self, rows_list, norms) -> bool:
        """May this attention's input projection run with RMSNorm + RoPE + the head-major re-layout fused into the GEMM epilogue (ST355_EPI_QK_NORM_ROPE,
        the form the fused processors name: flux/transformer.py:140-207)?  Built for head_dim 128 with an even head count and tile-aligned streams;
        the backward then recovers the normalised activations from the roped Q / K (x_hat = R^T z / w), which needs non-zero norm weights — checked here
        per call on the tiny [128] weight vectors of a frozen-base run (cached: they do not train under LoRA).  ST355_FUSED_QKV=0 keeps the separate pass."""
        if not _FUSED_QKV or self.hd != 128 or self.H % 2 or _TRANSPOSED_COPIES or any(r % 256 for r in rows_list):
            return False
        # keyed on identity AND the tensors' in-place version counters / requires_grad: weights loaded in place later (load_state_dict, copy_), a later
        # requires_grad_(True) or an id reused after a free can never hit a stale verdict; no device read on the hit path
        key = tuple((id(w), w._version, w.requires_grad, w.data_ptr()) if w is not None else None for w in norms)
        ok = self._norm_w_ok.get(key)
        if ok is None:
            ok = all(w is None or (not w.requires_grad and float(w.detach().abs().min()) > 1e-3) for w in norms)
            self._norm_w_ok[key] = ok
        return ok

    def _qk_fwd(self, env, qkv, wq, wk, Q, K, Qt, Kt, Vt, rows: int, pos0: int):
        """RMSNorm + RoPE + head-major re-layout of one stream's q / k / v rows (the separate pass of the unfused projection).  Under TREAD routing every
        sample carries its OWN position table (the kept image tokens' positions, flux/transformer.py:1211-1241 `_route_rope`): one call per sample."""
        B, H, hd, S, Sp = env.B, self.H, self.hd, env.S, env.Sp
        if not getattr(env, "routed", False):
            ops.qk_norm_rope_fwd(qkv, wq, wk, env.cos, env.sin, Q, K, Qt, Kt, Vt, B, H, hd, rows, pos0, S, Sp)
            return
        for b in range(B):
            ops.qk_norm_rope_fwd(qkv[b * S:(b + 1) * S], wq, wk, env.cos_b[b], env.sin_b[b], Q[b:b + 1], K[b:b + 1],
                                 None if Qt is None else Qt[b:b + 1], None if Kt is None else Kt[b:b + 1], Vt[b:b + 1], 1, H, hd, rows, pos0, S, Sp)

    def _attn_forward_fused(self, Q, K, V, Vt, O, lse2, env):
        """attention after the fused projection: from the head-major V^T the epilogue also wrote (default), or — ST355_FUSED_VT=0 — straight from the
        row-major V by transposing LDS reads (no V^T at all; measured r02: the forward kernel is ~4-8 % slower that way, the extra V^T write is cheaper)"""
        B, H, hd, S = env.B, self.H, self.hd, env.S
        if Vt is not None:
            ops.attn_fwd(Q, K, Vt, O, lse2, B, H, S, S, hd, env.scale, key_bias=env.key_bias)
        else:
            ops.attn_fwd_vrows(Q, K, V, O, lse2, B, H, S, hd, env.scale, key_bias=env.key_bias)

    def _alloc_heads(self, env):
        B, H, hd, S, Sp, dev = env.B, self.H, self.hd, env.S, env.Sp, self.device_
        Q = torch.empty(B, H, S, hd, dtype=BF16, device=dev); K = torch.empty_like(Q)
        mk = torch.zeros if Sp > S else torch.empty
        Vt = mk(B, H, hd, Sp, dtype=BF16, device=dev)
        Qt = Kt = None
        if hd != 128 or _TRANSPOSED_COPIES:
            # head_dim 64: the backward kernels read Q^T / K^T from pre-transposed head-major copies.  head_dim 128 (Flux.1): they gather those
            # fragments from the row-major tiles with transposing LDS reads (attention_bwd.hip dkv3 / dq<TR>) — two of the five head-major buffers and
            # their HBM passes are gone
            Qt = mk(B, H, hd, Sp, dtype=BF16, device=dev); Kt = mk(B, H, hd, Sp, dtype=BF16, device=dev)
        return Q, K, Qt, Kt, Vt

    def _double_fwd(self, bi: int, img, txt, env, save: bool):
        """one FluxTransformerBlock (flux/transformer.py:607-687).  The img (4096-row) and txt (512-row) streams use different weights but the same
        epilogues: every pair of projections goes out as ONE grouped launch, so the txt tiles fill the tail wave of the img GEMM.
        Returns (img', txt', x, saved): the LAST double block writes the joint [txt || img] sequence x of the single blocks in place instead."""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        B, Si, St, S, Sp, mod, cos, sin = env.B, env.Si, env.St, env.S, env.Sp, env.mod, env.cos, env.sin
        blk = self.double[bi]
        # the image stream's modulation rows: per sample, or — tokenwise timesteps (_engine_forward) — per image token (rows_per_batch 1); the text stream's stay per sample
        tokw = getattr(env, "tokenwise", False)
        mi = (env.mod_img if tokw else mod)[:, blk.mod_off:blk.mod_off + 6 * D]; mt = mod[:, blk.mod_off_c:blk.mod_off_c + 6 * D]
        rpi = 1 if tokw else Si
        full = getattr(env, "full", False)        # full-rank training: norm weights train (no fused projection epilogue), extra activations are kept
        fused = (not full) and (not getattr(env, "routed", False)) and self._fused_qkv_ok((Si, St), (blk.norm_q, blk.norm_k, blk.norm_added_q, blk.norm_added_k))
        ff_lora = any(l.lora is not None for l in (blk.ff1, blk.ff2, blk.ffc1, blk.ffc2))      # '+ffs' adapters: host sequencing
        keep_y = (full or self.mod_lora is not None) and save        # the un-gated branch outputs: gate gradients (full-rank training; adapters on the modulation Linears)
        if (fused and not tokw and not ff_lora and self.mod_lora is None and _block_abi_ok() and _BLOCK_ABI_ONLY in ("", "double", "fwd") and _FUSED_VT and blk.add_qkv.lora is None and blk.to_add_out.lora is None and img.is_contiguous() and txt.is_contiguous()):
            # the production form of the block as ONE C entry point (st355_block_flux_double_fwd): the same launches on the same operands as
            # the host-side sequencing below (adapters on the image stream's to_q / to_k / to_v / to_out.0, the reference's default target set)
            mk = lambda r, c: torch.empty(r, c, dtype=BF16, device=dev)
            lq, lo_ = blk.qkv.lora, blk.to_out.lora
            n_img, n_txt = mk(B * Si, D), mk(B * St, D)
            T_img = mk(B * Si, lq.K2) if lq is not None else None
            last_blk = bi == len(self.double) - 1
            Q = torch.empty(B, H, S, hd, dtype=BF16, device=dev); K = torch.empty_like(Q); Vt = torch.empty(B, H, hd, S, dtype=BF16, device=dev)
            V, O = mk(B * S, D), mk(B * S, D)
            rrms = torch.empty(B * S, 2 * H, dtype=F32, device=dev); lse2 = torch.empty(B, H, S, dtype=F32, device=dev)
            x1_img, x1_txt = mk(B * Si, D), mk(B * St, D)
            hpre_img, hpre_txt = mk(B * Si, 4 * D), mk(B * St, 4 * D)
            T_o = mk(B * Si, lo_.K2) if lo_ is not None else None
            x = mk(B * S, D) if last_blk else None
            x2_img, x2_txt = (None, None) if last_blk else (mk(B * Si, D), mk(B * St, D))
            ops.block_flux_double_fwd(B=B, Si=Si, St=St, H=H, D=D, K2_qkv=lq.K2 if lq is not None else 0, k2r_qkv=lq.k2_real if lq is not None else 0,
                                      K2_out=lo_.K2 if lo_ is not None else 0, k2r_out=lo_.k2_real if lo_ is not None else 0, scale=env.scale, img=img, txt=txt,
                                      mod_img=mi, mod_txt=mt, mod_stride=mod.stride(0), w_qkv=blk.qkv.w, b_qkv=blk.qkv.b, w_add_qkv=blk.add_qkv.w, b_add_qkv=blk.add_qkv.b,
                                      A_qkv=lq.A_cat if lq is not None else None, Bb_qkv=lq.B_blk if lq is not None else None,
                                      A_out=lo_.A_cat if lo_ is not None else None, Bb_out=lo_.B_blk if lo_ is not None else None,
                                      norm_q=blk.norm_q, norm_k=blk.norm_k, norm_added_q=blk.norm_added_q, norm_added_k=blk.norm_added_k,
                                      w_out=blk.to_out.w, b_out=blk.to_out.b, w_add_out=blk.to_add_out.w, b_add_out=blk.to_add_out.b,
                                      w_ff1=blk.ff1.w, b_ff1=blk.ff1.b, w_ff2=blk.ff2.w, b_ff2=blk.ff2.b, w_ffc1=blk.ffc1.w, b_ffc1=blk.ffc1.b, w_ffc2=blk.ffc2.w, b_ffc2=blk.ffc2.b,
                                      cos_p=env.cos_p, sin_p=env.sin_p, key_bias=env.key_bias, n_img=n_img, n_txt=n_txt, V=V, rrms=rrms, Q=Q, K=K, O=O, lse2=lse2,
                                      x1_img=x1_img, x1_txt=x1_txt, hpre_img=hpre_img, hpre_txt=hpre_txt, T_img=T_img, T_o=T_o, Vt=Vt,
                                      n2_img=mk(B * Si, D), n2_txt=mk(B * St, D), h_img=mk(B * Si, 4 * D), h_txt=mk(B * St, 4 * D),
                                      out_img=x2_img, out_txt=x2_txt, out_joint=x)
            sv = None
            if save:
                sv = SimpleNamespace(img=img, txt=txt, n_img=n_img, n_txt=None, qkv=None, V=V, rrms=rrms, Q=Q, K=K, Qt=None, Kt=None, O=O, lse2=lse2, x1_img=x1_img,
                                     x1_txt=x1_txt, hpre_img=hpre_img, hpre_txt=hpre_txt, T_img=T_img, T_txt=None, T_o=T_o, T_ao=None)
            return x2_img, x2_txt, x, sv
        n_img = ops.ln_modulate_fwd(img, mi[:, D:2 * D], mi[:, :D], rpi)
        n_txt = ops.ln_modulate_fwd(txt, mt[:, D:2 * D], mt[:, :D], St)
        T_img = ops.gemm(n_img, blk.qkv.lora.A_cat) if blk.qkv.lora is not None else None
        T_txt = ops.gemm(n_txt, blk.add_qkv.lora.A_cat) if blk.add_qkv.lora is not None else None
        kw_t = dict(a2=T_txt, b2=blk.add_qkv.lora.B_blk, k2_real=blk.add_qkv.lora.k2_real) if T_txt is not None else {}
        kw_i = dict(a2=T_img, b2=blk.qkv.lora.B_blk, k2_real=blk.qkv.lora.k2_real) if T_img is not None else {}
        O = torch.empty(B * S, D, dtype=BF16, device=dev); lse2 = torch.empty(B, H, S, dtype=F32, device=dev)
        qkv = V = rrms = Qt = Kt = None
        if fused:
            # RMSNorm(q), RMSNorm(k), RoPE and the head-major re-layout ride in the projection's epilogue: q / k leave the GEMM as the roped head-major
            # Q / K of the joint sequence, v as rows of the joint V (read row-major by the attention kernels), 1/rms for the backward — one launch for
            # both streams, no [tokens, 3D] intermediate, no pass over it
            Q = torch.empty(B, H, S, hd, dtype=BF16, device=dev); K = torch.empty_like(Q)
            V = torch.empty(B * S, D, dtype=BF16, device=dev); rrms = torch.empty(B * S, 2 * H, dtype=F32, device=dev)
            Vt = torch.empty(B, H, hd, S, dtype=BF16, device=dev) if _FUSED_VT else None      # (S is a multiple of 256 here: no padded columns)
            ops.gemm_grouped(self._problems(env, Si, dict(a=n_img, w=blk.qkv.w, bias=blk.qkv.b, out=self._rows_of(V, St, Si, env), epilogue=EPI_QK_NORM_ROPE,
                                                          rope=ops.qk_rope(Q, K, rrms, blk.norm_q, blk.norm_k, env.cos_p, env.sin_p, H, S, St, Vt=Vt),
                                                          rows_per_batch=Si, **kw_i))
                             + self._problems(env, St, dict(a=n_txt, w=blk.add_qkv.w, bias=blk.add_qkv.b, out=self._rows_of(V, 0, St, env),
                                                            epilogue=EPI_QK_NORM_ROPE, rows_per_batch=St,
                                                            rope=ops.qk_rope(Q, K, rrms, blk.norm_added_q, blk.norm_added_k, env.cos_p, env.sin_p, H, S, 0, Vt=Vt),
                                                            **kw_t)))
            self._attn_forward_fused(Q, K, V, Vt, O, lse2, env)
            del Vt
        else:
            qkv = torch.empty(B * S, 3 * D, dtype=BF16, device=dev)
            # both streams project into the joint [txt || img] rows of qkv (flux/transformer.py:171-190 concatenates q, k, v of the two streams)
            ops.gemm_grouped(self._problems(env, Si, dict(a=n_img, w=blk.qkv.w, bias=blk.qkv.b, out=self._rows_of(qkv, St, Si, env), **kw_i))
                             + self._problems(env, St, dict(a=n_txt, w=blk.add_qkv.w, bias=blk.add_qkv.b, out=self._rows_of(qkv, 0, St, env), **kw_t)))
            Q, K, Qt, Kt, Vt = self._alloc_heads(env)
            self._qk_fwd(env, qkv, blk.norm_added_q, blk.norm_added_k, Q, K, Qt, Kt, Vt, St, 0)
            self._qk_fwd(env, qkv, blk.norm_q, blk.norm_k, Q, K, Qt, Kt, Vt, Si, St)
            ops.attn_fwd(Q, K, Vt, O, lse2, B, H, S, Sp, hd, env.scale, key_bias=env.key_bias)
            del Vt
        x1_img = torch.empty(B * Si, D, dtype=BF16, device=dev); x1_txt = torch.empty(B * St, D, dtype=BF16, device=dev)
        T_o = torch.empty(B * Si, blk.to_out.lora.K2, dtype=BF16, device=dev) if blk.to_out.lora is not None else None
        T_ao = torch.empty(B * St, blk.to_add_out.lora.K2, dtype=BF16, device=dev) if blk.to_add_out.lora is not None else None
        O_i, O_t = self._rows_of(O, St, Si, env), self._rows_of(O, 0, St, env)       # the attention output is split back by rows, in place
        kw_i, kw_t = {}, {}
        ya_i = ya_t = yf_i = yf_t = None
        if keep_y:                   # the un-gated branch outputs (gate gradients: d gate = sum_rows dOut * y)
            ya_i, yf_i = (torch.empty(B * Si, D, dtype=BF16, device=dev) for _ in range(2))
            ya_t, yf_t = (torch.empty(B * St, D, dtype=BF16, device=dev) for _ in range(2))
            kw_i["aux_out"], kw_t["aux_out"] = ya_i, ya_t
        if T_o is not None:
            for pr in self._problems(env, Si, dict(a=O_i, w=blk.to_out.lora.A_cat, out=T_o)):
                ops.gemm(pr.pop("a"), pr.pop("w"), **pr)
            kw_i.update(a2=T_o, b2=blk.to_out.lora.B_blk, k2_real=blk.to_out.lora.k2_real)
        if T_ao is not None:
            for pr in self._problems(env, St, dict(a=O_t, w=blk.to_add_out.lora.A_cat, out=T_ao)):
                ops.gemm(pr.pop("a"), pr.pop("w"), **pr)
            kw_t.update(a2=T_ao, b2=blk.to_add_out.lora.B_blk, k2_real=blk.to_add_out.lora.k2_real)
        ops.gemm_grouped(self._problems(env, Si, dict(a=O_i, w=blk.to_out.w, bias=blk.to_out.b, out=x1_img, epilogue=EPI_GATE_RESIDUAL, aux_in=img,
                                                      gate=mi[:, 2 * D:3 * D], rows_per_batch=rpi, **kw_i))
                         + self._problems(env, St, dict(a=O_t, w=blk.to_add_out.w, bias=blk.to_add_out.b, out=x1_txt, epilogue=EPI_GATE_RESIDUAL,
                                                        aux_in=txt, gate=mt[:, 2 * D:3 * D], rows_per_batch=St, **kw_t)))
        # MLPs
        n2_i = ops.ln_modulate_fwd(x1_img, mi[:, 4 * D:5 * D], mi[:, 3 * D:4 * D], rpi)
        n2_t = ops.ln_modulate_fwd(x1_txt, mt[:, 4 * D:5 * D], mt[:, 3 * D:4 * D], St)
        hpre_img = torch.empty(B * Si, 4 * D, dtype=BF16, device=dev); hpre_txt = torch.empty(B * St, 4 * D, dtype=BF16, device=dev)
        # '+ffs' adapters (flux/model.py:1272-1301): the same K-extension as the attention projections' — T = x A^T first, then [x | T] [W | sB]^T in the projection's launch
        ext = lambda lg, T_: dict(a2=T_, b2=lg.B_blk, k2_real=lg.k2_real) if lg is not None else {}
        T_f1 = ops.gemm(n2_i, blk.ff1.lora.A_cat) if blk.ff1.lora is not None else None
        T_c1 = ops.gemm(n2_t, blk.ffc1.lora.A_cat) if blk.ffc1.lora is not None else None
        h_i, h_t = ops.gemm_grouped([dict(a=n2_i, w=blk.ff1.w, bias=blk.ff1.b, epilogue=EPI_GELU, aux_out=hpre_img, **ext(blk.ff1.lora, T_f1)),
                                     dict(a=n2_t, w=blk.ffc1.w, bias=blk.ffc1.b, epilogue=EPI_GELU, aux_out=hpre_txt, **ext(blk.ffc1.lora, T_c1))])
        T_f2 = ops.gemm(h_i, blk.ff2.lora.A_cat) if blk.ff2.lora is not None else None
        T_c2 = ops.gemm(h_t, blk.ffc2.lora.A_cat) if blk.ffc2.lora is not None else None
        x = x2_img = x2_txt = None
        kf_i = dict(aux_out=yf_i) if yf_i is not None else {}
        kf_t = dict(aux_out=yf_t) if yf_t is not None else {}
        kf_i.update(ext(blk.ff2.lora, T_f2)); kf_t.update(ext(blk.ffc2.lora, T_c2))
        if bi == len(self.double) - 1:
            # the last double block's MLP down-projections write the joint [txt || img] sequence of the single blocks in place
            # (flux/transformer.py:1332 `torch.cat`): one problem per (stream, sample), no concat pass
            x = torch.empty(B * S, D, dtype=BF16, device=dev)
            ops.gemm_grouped(self._problems(env, Si, dict(a=h_i, w=blk.ff2.w, bias=blk.ff2.b, epilogue=EPI_GATE_RESIDUAL, aux_in=x1_img,
                                                          gate=mi[:, 5 * D:6 * D], rows_per_batch=rpi, out=self._rows_of(x, St, Si, env), **kf_i))
                             + self._problems(env, St, dict(a=h_t, w=blk.ffc2.w, bias=blk.ffc2.b, epilogue=EPI_GATE_RESIDUAL, aux_in=x1_txt,
                                                            gate=mt[:, 5 * D:6 * D], rows_per_batch=St, out=self._rows_of(x, 0, St, env), **kf_t)))
        else:
            x2_img, x2_txt = ops.gemm_grouped([
                dict(a=h_i, w=blk.ff2.w, bias=blk.ff2.b, epilogue=EPI_GATE_RESIDUAL, aux_in=x1_img, gate=mi[:, 5 * D:6 * D], rows_per_batch=rpi, **kf_i),
                dict(a=h_t, w=blk.ffc2.w, bias=blk.ffc2.b, epilogue=EPI_GATE_RESIDUAL, aux_in=x1_txt, gate=mt[:, 5 * D:6 * D], rows_per_batch=St, **kf_t)])
        sv = None
        if save:
            sv = SimpleNamespace(img=img, txt=txt, n_img=n_img, n_txt=n_txt if (T_txt is not None or full) else None, qkv=qkv, V=V, rrms=rrms, Q=Q, K=K, Qt=Qt, Kt=Kt, O=O,
                                 lse2=lse2, x1_img=x1_img, x1_txt=x1_txt, hpre_img=hpre_img, hpre_txt=hpre_txt, T_img=T_img, T_txt=T_txt, T_o=T_o, T_ao=T_ao)
            if full:    # full-rank training also needs every Linear's input (weight gradients) and the un-gated branch outputs (gate gradients)
                sv.n2_i, sv.n2_t, sv.h_i, sv.h_t, sv.ya_i, sv.ya_t, sv.yf_i, sv.yf_t = n2_i, n2_t, h_i, h_t, ya_i, ya_t, yf_i, yf_t
            elif keep_y:
                sv.ya_i, sv.ya_t, sv.yf_i, sv.yf_t = ya_i, ya_t, yf_i, yf_t
            if ff_lora:  # the feed-forward adapters' gradients read their Linear's input (dA = U^T x) and T = x A^T (dB = s dy^T T)
                sv.ff = SimpleNamespace(n2_i=n2_i if blk.ff1.lora is not None else None, n2_t=n2_t if blk.ffc1.lora is not None else None,
                                        h_i=h_i if blk.ff2.lora is not None else None, h_t=h_t if blk.ffc2.lora is not None else None,
                                        T_f1=T_f1, T_c1=T_c1, T_f2=T_f2, T_c2=T_c2)
        return x2_img, x2_txt, x, sv

    def _single_fwd(self, bi: int, x, env, save: bool):
        """one FluxSingleTransformerBlock (flux/transformer.py:473-510)"""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        B, S, Sp, mod, cos, sin = env.B, env.S, env.Sp, env.mod, env.cos, env.sin
        blk = self.single[bi]
        tokw = getattr(env, "tokenwise", False)       # tokenwise timesteps: one modulation row per token of the joint [txt || img] sequence (env.mod_x: columns from env.xoff on)
        ms = env.mod_x[:, blk.mod_off - env.xoff:blk.mod_off - env.xoff + 3 * D] if tokw else mod[:, blk.mod_off:blk.mod_off + 3 * D]
        rpx = 1 if tokw else S
        full = getattr(env, "full", False)
        fused = (not full) and (not getattr(env, "routed", False)) and self._fused_qkv_ok((S,), (blk.norm_q, blk.norm_k))
        ff_lora = blk.proj_mlp.lora is not None or blk.proj_out.lora is not None      # '+ffs' / 'tiny' / 'nano' adapters: host sequencing (the C entry point knows the q / k / v adapters)
        keep_y = (full or self.mod_lora is not None) and save
        if fused and not tokw and not ff_lora and self.mod_lora is None and _block_abi_ok() and _BLOCK_ABI_ONLY in ("", "single", "fwd") and _FUSED_VT and x.is_contiguous():
            # the production form of the block as ONE C entry point (st355_block_flux_single_fwd): the same launches on the same operands
            # as the host-side sequencing below
            lo = blk.qkv.lora
            n = torch.empty(B * S, D, dtype=BF16, device=dev); O = torch.empty_like(n); V = torch.empty_like(n); x_out = torch.empty_like(n)
            Q = torch.empty(B, H, S, hd, dtype=BF16, device=dev); K = torch.empty_like(Q); Vt = torch.empty(B, H, hd, S, dtype=BF16, device=dev)
            rrms = torch.empty(B * S, 2 * H, dtype=F32, device=dev); lse2 = torch.empty(B, H, S, dtype=F32, device=dev)
            hpre = torch.empty(B * S, 4 * D, dtype=BF16, device=dev); hact = torch.empty_like(hpre)
            T = torch.empty(B * S, lo.K2, dtype=BF16, device=dev) if lo is not None else None
            ops.block_flux_single_fwd(B=B, S=S, H=H, D=D, K2=lo.K2 if lo is not None else 0, k2_real=lo.k2_real if lo is not None else 0, scale=env.scale,
                                      x=x, mod_shift=ms[:, :D], mod_scale=ms[:, D:2 * D], mod_gate=ms[:, 2 * D:3 * D], mod_stride=ms.stride(0),
                                      w_qkv=blk.qkv.w, b_qkv=blk.qkv.b, A_cat=lo.A_cat if lo is not None else None, B_blk=lo.B_blk if lo is not None else None,
                                      norm_q=blk.norm_q, norm_k=blk.norm_k, w_mlp=blk.proj_mlp.w, b_mlp=blk.proj_mlp.b,
                                      w_out=blk.proj_out.w, ld_w_out=blk.proj_out.w.stride(0), b_out=blk.proj_out.b, cos_p=env.cos_p, sin_p=env.sin_p,
                                      key_bias=env.key_bias, n=n, V=V, rrms=rrms, Q=Q, K=K, O=O, lse2=lse2, hpre=hpre, T=T, Vt=Vt, hact=hact, x_out=x_out)
            sv = SimpleNamespace(x=x, n=n, qkv=None, V=V, rrms=rrms, Q=Q, K=K, Qt=None, Kt=None, O=O, lse2=lse2, hpre=hpre, T=T, T_m=None, T_p=None) if save else None
            return x_out, sv
        n = ops.ln_modulate_fwd(x, ms[:, D:2 * D], ms[:, :D], rpx)
        O = torch.empty(B * S, D, dtype=BF16, device=dev); lse2 = torch.empty(B, H, S, dtype=F32, device=dev)
        qkv = V = rrms = Qt = Kt = None
        if fused:
            Q = torch.empty(B, H, S, hd, dtype=BF16, device=dev); K = torch.empty_like(Q)
            V = torch.empty(B * S, D, dtype=BF16, device=dev); rrms = torch.empty(B * S, 2 * H, dtype=F32, device=dev)
            Vt = torch.empty(B, H, hd, S, dtype=BF16, device=dev) if _FUSED_VT else None
            _, T = self._lin_fwd(blk.qkv, n, out=V, epilogue=EPI_QK_NORM_ROPE, rows_per_batch=S,
                                 rope=ops.qk_rope(Q, K, rrms, blk.norm_q, blk.norm_k, env.cos_p, env.sin_p, H, S, 0, Vt=Vt))
            self._attn_forward_fused(Q, K, V, Vt, O, lse2, env)
            del Vt
        else:
            qkv, T = self._lin_fwd(blk.qkv, n)
            Q, K, Qt, Kt, Vt = self._alloc_heads(env)
            self._qk_fwd(env, qkv, blk.norm_q, blk.norm_k, Q, K, Qt, Kt, Vt, S, 0)
            ops.attn_fwd(Q, K, Vt, O, lse2, B, H, S, Sp, hd, env.scale, key_bias=env.key_bias)
            del Vt
        hpre = torch.empty(B * S, 4 * D, dtype=BF16, device=dev)
        lm, lp = blk.proj_mlp.lora, blk.proj_out.lora
        T_m = ops.gemm(n, lm.A_cat) if lm is not None else None
        hact = ops.gemm(n, blk.proj_mlp.w, bias=blk.proj_mlp.b, epilogue=EPI_GELU, aux_out=hpre,
                        **(dict(a2=T_m, b2=lm.B_blk, k2_real=lm.k2_real) if lm is not None else {}))
        # cat[attn, mlp] @ Wout^T is a two-segment K loop: no [B,S,5D] concat buffer is ever materialised
        y = torch.empty(B * S, D, dtype=BF16, device=dev) if keep_y else None          # the un-gated branch output (gate gradient)
        x_in, T_p = x, None
        if lp is not None:
            # proj_out's adapter: T = [attn | mlp] A^T as the same two-segment K loop; both K segments of the projection are taken, so the low-rank term goes
            # out as its own gated-residual launch first:  x' = x + gate * (T (sB)^T),  then  x_out = x' + gate * ([attn | mlp] W^T + b)
            T_p = ops.gemm(O, lp.A_cat[:, :D], a2=hact, b2=lp.A_cat[:, D:])
            x_in = ops.gemm(T_p, lp.B_blk, epilogue=EPI_GATE_RESIDUAL, aux_in=x, gate=ms[:, 2 * D:3 * D], rows_per_batch=rpx)
        x_out = ops.gemm(O, blk.proj_out.w[:, :D], bias=blk.proj_out.b, a2=hact, b2=blk.proj_out.w[:, D:], epilogue=EPI_GATE_RESIDUAL,
                         aux_in=x_in, gate=ms[:, 2 * D:3 * D], rows_per_batch=rpx, **(dict(aux_out=y) if y is not None else {}))
        if y is not None and lp is not None:          # the gate gradient wants the WHOLE un-gated branch: the projection's output + its adapter's low-rank term
            y = ops.gemm(T_p, lp.B_blk, epilogue=EPI_ADD, aux_in=y)
        sv = SimpleNamespace(x=x, n=n, qkv=qkv, V=V, rrms=rrms, Q=Q, K=K, Qt=Qt, Kt=Kt, O=O, lse2=lse2, hpre=hpre, T=T, T_m=T_m, T_p=T_p) if save else None
        if sv is not None and (full or lp is not None or keep_y):
            sv.hact, sv.y = hact, y           # (proj_out's adapter gradient dA = U^T [attn | mlp] reads the activated MLP rows)
        return x_out, sv

    def _engine_forward(self, hidden_states, encoder_hidden_states, pooled, timestep, guidance, img_ids, txt_ids, save: bool, key_bias=None, full: bool = False):
        D, H, hd = self.D, self.H, self.hd
        B, Si, _ = hidden_states.shape
        St = encoder_hidden_states.shape[1]
        S = Si + St
        Sp = (S + 63) // 64 * 64
        dev = self.device_
        for g in self.lora_groups:
            g.pack()
        cos, sin, cos_p, sin_p = self._rope(txt_ids, img_ids)
        # ---- embeddings (flux/transformer.py:1001-1064) ----
        em = SimpleNamespace()        # the embedders' intermediates (kept for their weight gradients under full-rank training)
        em.x2d, em.enc2d = hidden_states.reshape(B * Si, -1).contiguous(), encoder_hidden_states.reshape(B * St, -1).contiguous()
        img, T_x = self._lin_fwd(self.l_x, em.x2d)          # ('all+ffs+embedder' wraps x_embedder)
        txt = ops.gemm(em.enc2d, self.l_ctx.w, bias=self.l_ctx.b)
        tokenwise = timestep.dim() == 2
        if tokenwise and tuple(timestep.shape) != (B, Si):
            raise ValueError(f"Flux expected tokenwise timesteps with sequence length {Si}, got {timestep.shape[1]}.")     # flux/transformer.py:1068-1072
        if self.mod_lora is not None and (tokenwise or full):
            raise NotImplementedError("flux_lora_target='ai-toolkit' (adapters on the AdaLN modulation Linears) takes per-sample timesteps under LoRA training on the st355 path")
        t32 = timestep.to(device=dev, dtype=F32).reshape(-1).contiguous()
        em.tproj = ops.timestep_proj(t32, 256, 1000.0)
        em.t1 = ops.gemm(em.tproj, self.l_t1.w, bias=self.l_t1.b); em.st1 = ops.silu(em.t1)
        temb = ops.gemm(em.st1, self.l_t2.w, bias=self.l_t2.b)                   # [B, D], or tokenwise [B * S_img, D]
        cond = None                                                              # the per-SAMPLE part of the conditioning: guidance + pooled text
        if self.config.guidance_embeds:
            if guidance is None:
                raise ValueError("guidance_embeds=True requires a guidance tensor")
            g32 = guidance.to(device=dev, dtype=F32).contiguous()
            if tokenwise and g32.dim() != 1:
                raise NotImplementedError("tokenwise timesteps take a per-sample guidance vector on the st355 path (a [B, S] guidance tensor is not implemented)")
            em.gproj = ops.timestep_proj(g32, 256, 1000.0)
            em.g1 = ops.gemm(em.gproj, self.l_g1.w, bias=self.l_g1.b); em.sg1 = ops.silu(em.g1)
            cond = ops.gemm(em.sg1, self.l_g2.w, bias=self.l_g2.b)
        em.pooled = pooled.to(BF16).contiguous()
        em.p1 = ops.gemm(em.pooled, self.l_p1.w, bias=self.l_p1.b); em.sp1 = ops.silu(em.p1)
        pe = ops.gemm(em.sp1, self.l_p2.w, bias=self.l_p2.b)
        mod_img = mod_x = None
        xoff = 0
        if tokenwise:
            # TOKENWISE timesteps [B, S_img] (CREPA self-flow; flux/transformer.py:245-294, 386-412, 1068-1086, 1505): one conditioning row per image token.  The image
            # stream of the double blocks and norm_out take per-token shift / scale / gate rows (the AdaLN / gated-residual / scale kernels index their modulation row
            # by row // rows_per_batch: rows_per_batch = 1), the text stream the mean over the tokens, the single blocks [mean x S_txt || per token] along their joint
            # sequence.  The per-token rows of every block are ONE GEMM [B * S_img, D] x [mod_total, D]^T (B * S_img * mod_total bf16: 8.7 GB per 1024^2 image of
            # Flux.1 — HBM holds it); the joint-sequence rows of the single blocks + norm_out are assembled from them and the per-sample rows.
            if full:
                raise NotImplementedError("tokenwise timesteps under full-rank training (per-token modulation gradients) are not implemented on the st355 path")
            # the three embedders' sum, fp32 accumulation order of the batch-wise path: timestep + guidance, then + pooled
            if cond is not None:
                temb = ops.add(temb, cond[:, None, :].expand(B, Si, D).reshape(B * Si, D))
            temb_tok = ops.add(temb, pe[:, None, :].expand(B, Si, D).reshape(B * Si, D))
            tsum = torch.empty(B, D, dtype=F32, device=dev)
            ops.colsum_prod(temb_tok, tsum, rows_per_batch=Si)
            temb = (tsum / Si).to(BF16)                                                               # temb_txt = temb.mean(dim=1)   (:1073-1074)
            mod_img = ops.gemm(ops.silu(temb_tok), self.mod_w, bias=self.mod_b)                       # [B * S_img, mod_total]
        else:
            if cond is not None:
                temb = ops.add(temb, cond)
            temb = ops.add(temb, pe)
        em.temb, em.st = temb, ops.silu(temb)
        ml = self.mod_lora
        T_mod = ops.gemm(em.st, ml.A_cat) if ml is not None else None
        mod = ops.gemm(em.st, self.mod_w, bias=self.mod_b, **(dict(a2=T_mod, b2=ml.B_blk, k2_real=ml.k2_real) if ml is not None else {}))   # [B, mod_total]: every block's modulation at once
        if tokenwise:
            xoff = self.single[0].mod_off if self.single else self.mod_off_out          # the single blocks' and norm_out's columns follow the double blocks'
            mod_x = torch.cat([mod[:, None, xoff:].expand(B, St, self.mod_total - xoff), mod_img.view(B, Si, -1)[:, :, xoff:]], dim=1).reshape(B * S, self.mod_total - xoff)
        env = SimpleNamespace(B=B, Si=Si, St=St, S=S, Sp=Sp, cos=cos, sin=sin, cos_p=cos_p, sin_p=sin_p, mod=mod, scale=1.0 / math.sqrt(hd), key_bias=key_bias,
                              full=bool(full), tokenwise=tokenwise, mod_img=mod_img, mod_x=mod_x, xoff=xoff)
        segs_d = self._checkpoint_segments(len(self.double)) if save else [(i, 1, False) for i in range(len(self.double))]
        segs_s = self._checkpoint_segments(len(self.single)) if save else [(i, 1, False) for i in range(len(self.single))]
        # TREAD routing (flux/transformer.py:1101-1133, 1211-1241 double blocks, 1394-1486 single blocks; training/tread.py): only while training; between a route's
        # two blocks the IMAGE stream is a per-sample subset of its tokens, the text tokens all stay.  Layer indices run over double + single blocks.  Under
        # routing the reference drops the segmented checkpoint form for the per-block one (:1144-1157 `not use_routing`).
        from ..training.tread import normalise_routes
        nd, ns = len(self.double), len(self.single)
        routes = normalise_routes(self._tread_routes, nd + ns) if (save and self.training and self._tread_router is not None) else []
        if routes and tokenwise:
            raise NotImplementedError("tokenwise timesteps under TREAD routing (the routed tokens' modulation rows would be gathered too) are not implemented")
        if routes and ml is not None:
            raise NotImplementedError("TREAD routing with flux_lora_target='ai-toolkit' (modulation-row gradients over routed token subsets) is not built on the st355 path")
        if routes and full:
            raise NotImplementedError("TREAD routing under full-rank Flux training is not built on the st355 path (LoRA training routes)")
        if routes:
            from ..training.checkpoint_plan import per_block as _per_block
            gc_, iv_, sd_ = bool(self.gradient_checkpointing), self.gradient_checkpointing_interval, self.gradient_checkpointing_segment_stride
            segs_d, segs_s = _per_block(nd, gc_, iv_, sd_), _per_block(ns, gc_, iv_, sd_)
        ctx = SimpleNamespace(env=env, dbl={}, sgl={}, segs_d=segs_d, segs_s=segs_s, ck_d={}, ck_s={}, env_d=[env] * nd, env_s=[env] * ns,
                              route_start={}, route_end={})
        rt = SimpleNamespace(ptr=0, info=None, saved=None, env=env)          # the open route: its mask, the full image stream at its start, the routed env

        def start_route(img_full, g):
            """img_full [B*Si, D] (contiguous) -> the kept tokens [B*K, D]; opens the route"""
            info = self._tread_router.get_mask(img_full.view(B, Si, D), mask_ratio=routes[rt.ptr]["selection_ratio"], force_keep=getattr(self, "_force_keep_mask", None))
            K = info.ids_keep.shape[1]
            idx = info.ids_keep.to(dev)
            # position tables [text | kept image tokens] per sample (`_route_rope`, flux/transformer.py:909-938)
            cos_b = torch.cat([cos[:St][None].expand(B, -1, -1), cos[St:][idx]], dim=1).contiguous()
            sin_b = torch.cat([sin[:St][None].expand(B, -1, -1), sin[St:][idx]], dim=1).contiguous()
            kb = None if key_bias is None else key_bias[:, :St + K].contiguous()          # image keys are never masked (expand_flux_attention_mask)
            rt.info, rt.saved = info, img_full
            rt.env = SimpleNamespace(**{**vars(env), "Si": K, "S": St + K, "Sp": (St + K + 63) // 64 * 64, "routed": True, "cos_b": cos_b, "sin_b": sin_b, "key_bias": kb})
            ctx.route_start[g] = info
            return ops.gather_rows(img_full.view(B, Si, D), info.keep_i32()).view(-1, D)

        def end_route(img_r, g):
            """the processed kept tokens [B*K, D] back into their slots of the stream as it was at the route's start (TREADRouter.end_route)"""
            whole = rt.saved.clone()
            ops.scatter_rows(img_r.view(B, rt.env.Si, D), rt.info.keep_i32(), whole.view(B, Si, D))
            ctx.route_end[g] = rt.info
            rt.info, rt.saved, rt.env, rt.ptr = None, None, env, rt.ptr + 1
            return whole

        def starts(g):
            return rt.ptr < len(routes) and rt.info is None and g == routes[rt.ptr]["start_layer_idx"]

        def ends(g):
            return rt.info is not None and g == routes[rt.ptr]["end_layer_idx"]

        x = None
        # nothing below the first block that carries an adapter is differentiated (add_lora_adapter: 'tiny' / 'nano' start at single block 7): those blocks keep
        # no activations and no checkpoint inputs
        stop = 0 if full else getattr(self, "_bwd_stop", 0)
        # ---- double blocks ----
        for (s0, n, ck) in segs_d:
            for bi in range(s0, s0 + n):
                if starts(bi):
                    img = start_route(img, bi)
                if ck and bi == s0 and save and s0 + n - 1 >= stop:
                    ctx.ck_d[s0] = (img, txt)                 # a checkpointed segment keeps only its input; its blocks are re-run in backward
                ctx.env_d[bi] = rt.env
                img, txt, x, sv = self._double_fwd(bi, img, txt, rt.env, save and not ck and bi >= stop)
                if sv is not None:
                    ctx.dbl[bi] = sv
                if ends(bi):
                    if x is not None:                         # the last double block wrote the joint [txt || kept img] sequence: re-open it
                        xv = x.view(B, rt.env.S, D)
                        t_part = xv[:, :St]
                        restored = end_route(xv[:, St:].contiguous().view(-1, D), bi)
                        x = torch.cat([t_part, restored.view(B, Si, D)], dim=1).reshape(B * S, D)
                    else:
                        img = end_route(img, bi)
        # ---- joint sequence [txt || img] (flux/transformer.py:1332): written in place by the last double block ----
        if not self.double:
            x = torch.cat([txt.view(B, St, D), img.view(B, Si, D)], dim=1).reshape(B * S, D)
        del img, txt
        # ---- single blocks ----
        for (s0, n, ck) in segs_s:
            for bi in range(s0, s0 + n):
                g = nd + bi
                if starts(g):
                    xv = x.view(B, S, D)
                    t_part = xv[:, :St]
                    x = torch.cat([t_part, start_route(xv[:, St:].contiguous().view(-1, D), g).view(B, -1, D)], dim=1).reshape(-1, D)
                if ck and bi == s0 and save and nd + s0 + n - 1 >= stop:
                    ctx.ck_s[s0] = x
                ctx.env_s[bi] = rt.env
                x, sv = self._single_fwd(bi, x, rt.env, save and not ck and g >= stop)
                if sv is not None:
                    ctx.sgl[bi] = sv
                if ends(g):
                    xv = x.view(B, rt.env.S, D)
                    t_part = xv[:, :St]
                    restored = end_route(xv[:, St:].contiguous().view(-1, D), g)
                    x = torch.cat([t_part, restored.view(B, Si, D)], dim=1).reshape(B * S, D)
        if rt.info is not None:
            raise ValueError("TREAD route does not end inside the block stack (end_layer_idx)")
        # ---- output head (flux/transformer.py:1501-1506): AdaLayerNormContinuous chunk order is (scale, shift) ----
        mo = mod[:, self.mod_off_out:self.mod_off_out + 2 * D]
        n_out = torch.empty(B * Si, D, dtype=BF16, device=dev)
        for b in range(B):          # the image rows of sample b are a strided view of the joint buffer: no gather pass
            if tokenwise:           # per-token (scale, shift) rows: norm_out takes temb_img (flux/transformer.py:1505)
                mo_b = env.mod_img[b * Si:(b + 1) * Si, self.mod_off_out:self.mod_off_out + 2 * D]
                ops.ln_modulate_fwd(x[b * S + St:(b + 1) * S], mo_b[:, :D], mo_b[:, D:2 * D], 1, out=n_out[b * Si:(b + 1) * Si])
                continue
            ops.ln_modulate_fwd(x[b * S + St:(b + 1) * S], mo[b:b + 1, :D], mo[b:b + 1, D:2 * D], Si, out=n_out[b * Si:(b + 1) * Si])
        out, T_out = self._lin_fwd(self.l_out, n_out)          # ('all+ffs' also wraps the model's own proj_out: peft matches the name suffix, flux/model.py:1283-1301)
        if save:
            ctx.x_final = x
            if full:
                ctx.n_out, ctx.emb = n_out, em
            elif self.l_out.lora is not None:
                ctx.n_out, ctx.T_out = n_out, T_out
            if self.l_x.lora is not None:
                ctx.x2d, ctx.T_x = em.x2d, T_x
            if ml is not None:
                ctx.st, ctx.T_mod = em.st, T_mod
        return out.view(B, Si, -1), ctx

    def _attn_backward(self, sv, dO, dqkv, env):
        B, H, hd, S, Sp, D, dev = env.B, self.H, self.hd, env.S, env.Sp, self.D, self.device_
        dQ = torch.empty(B, H, S, hd, dtype=BF16, device=dev); dK = torch.empty_like(dQ)
        v_rows = sv.V if sv.V is not None else sv.qkv[:, 2 * D:]          # fused projection: V has its own [B*S, D] rows
        ops.attn_bwd(sv.Q, sv.K, sv.Qt, sv.Kt, v_rows, sv.O, dO, sv.lse2, dQ, dK, dqkv[:, 2 * D:], B, H, S, Sp, hd, env.scale, key_bias=env.key_bias)
        return dQ, dK

    def _attn_rope_backward(self, sv, dO, dqkv, env, w_lo, w_hi, split: int):
        """attention backward + RoPE / RMSNorm backward -> all three column blocks of dqkv.  After a fused projection (head_dim 128) the second half
        runs in the dQ / dK kernels' epilogues (one call, no head-major dQ / dK); otherwise attention backward, then the separate pass per stream:
        joint positions < split carry the text stream's norm weights w_lo = (q, k), the rest w_hi."""
        B, H, hd, S, Sp = env.B, self.H, self.hd, env.S, env.Sp
        if sv.rrms is not None and _FUSED_ROPE_BWD:
            ops.attn_bwd_rope(sv.Q, sv.K, sv.V, sv.O, dO, sv.lse2, sv.rrms, w_lo[0], w_lo[1], w_hi[0], w_hi[1], split, env.cos_p, env.sin_p, dqkv,
                              B, H, S, Sp, hd, env.scale, key_bias=env.key_bias)
            return
        dQ, dK = self._attn_backward(sv, dO, dqkv, env)
        if split > 0:
            self._rope_backward(sv, dQ, dK, w_lo[0], w_lo[1], dqkv, env, split, 0)
        self._rope_backward(sv, dQ, dK, w_hi[0], w_hi[1], dqkv, env, S - split, split)

    def _rope_backward(self, sv, dQ, dK, wq, wk, dqkv, env, rows, pos0):
        """dq, dk columns of dqkv for the `rows` tokens at joint position pos0: from the roped Q / K + 1/rms when the projection ran fused, else from
        the kept pre-norm projection"""
        B, H, hd, S = env.B, self.H, self.hd, env.S
        if sv.rrms is not None:
            ops.qk_rope_norm_bwd(dQ, dK, sv.Q, sv.K, sv.rrms, wq, wk, env.cos, env.sin, dqkv, B, H, hd, rows, pos0, S)
        elif getattr(env, "routed", False):         # TREAD: per-sample position tables (see _qk_fwd)
            for b in range(B):
                ops.qk_norm_rope_bwd(dQ[b:b + 1], dK[b:b + 1], sv.qkv[b * S:(b + 1) * S], wq, wk, env.cos_b[b], env.sin_b[b], dqkv[b * S:(b + 1) * S],
                                     1, H, hd, rows, pos0, S)
        else:
            ops.qk_norm_rope_bwd(dQ, dK, sv.qkv, wq, wk, env.cos, env.sin, dqkv, B, H, hd, rows, pos0, S)

    def _mod_grads(self, dn, x_in, rows: int, k_shift: int, k_scale: int, dm):
        """d shift = sum_rows dY, d scale = sum_rows dY * LN(x) of one AdaLN instance, per sample, into chunks k_shift / k_scale of its slice `dm` of the modulation-row
        gradient (the reductions of the full-rank engine's `mod_grads`; here for adapters on the modulation Linears under a frozen base)"""
        D = self.D
        ops.colsum_prod(dn, dm[:, k_shift * D:(k_shift + 1) * D], rows_per_batch=rows)
        ops.colsum_prod(dn, dm[:, k_scale * D:(k_scale + 1) * D], b=ops.layer_norm_xhat(x_in), rows_per_batch=rows)

    def _single_bwd(self, li: int, sv, dx, dxg, env):
        """backward of single block li.  Returns (dx, dxg, d_txt, d_img): block 0 of a model with double blocks writes its input gradient — the joint
        gradient of the double stack — per (stream, sample) straight into the two stream-major buffers (no split / gather pass)."""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        B, Si, St, S, mod, cos, sin = env.B, env.Si, env.St, env.S, env.mod, env.cos, env.sin
        blk = self.single[li]
        tokw = getattr(env, "tokenwise", False)       # one modulation row per token of the joint sequence (see _single_fwd)
        msl = (lambda j: env.mod_x[:, self.single[j].mod_off - env.xoff:self.single[j].mod_off - env.xoff + 3 * D]) if tokw else (lambda j: mod[:, self.single[j].mod_off:self.single[j].mod_off + 3 * D])
        ms = msl(li)
        rpx = 1 if tokw else S
        lm, lp = blk.proj_mlp.lora, blk.proj_out.lora
        dm = self._dmod[:, blk.mod_off:blk.mod_off + 3 * D] if self._dmod is not None else None      # 'ai-toolkit': this block's slice of d loss / d (modulation rows)
        if (not tokw and lm is None and lp is None and dm is None and _block_abi_ok() and _BLOCK_ABI_ONLY in ("", "single", "bwd") and _FUSED_ROPE_BWD and sv.rrms is not None and not getattr(env, "routed", False) and (li > 0 or not self.double)
                and dx.is_contiguous() and (dxg is None or dxg.is_contiguous())):
            # ONE C entry point (st355_block_flux_single_bwd): the launches of the host-side sequencing below, in its order, on its operands
            lo = blk.qkv.lora
            gprev = mod[:, self.single[li - 1].mod_off + 2 * D:self.single[li - 1].mod_off + 3 * D] if li > 0 else None
            mk = lambda c: torch.empty(B * S, c, dtype=BF16, device=dev)
            dx_out = mk(D)
            dxg_out = mk(D) if gprev is not None else None
            ops.block_flux_single_bwd([t for t in lo.gA] if lo is not None else None, [t for t in lo.gB] if lo is not None else None,
                                      B=B, S=S, H=H, D=D, K2=lo.K2 if lo is not None else 0, k2_real=lo.k2_real if lo is not None else 0,
                                      n_targets=len(lo.targets) if lo is not None else 0, rank=lo.rank if lo is not None else 0, r_pad=lo.r_pad if lo is not None else 0,
                                      accumulate=1 if self.accumulate_lora_grads else 0, scale=env.scale, lora_scale=lo.scale if lo is not None else 0.0,
                                      x=sv.x, n=sv.n, V=sv.V, rrms=sv.rrms, Q=sv.Q, K=sv.K, O=sv.O, lse2=sv.lse2, hpre=sv.hpre, T=sv.T,
                                      mod_scale=ms[:, D:2 * D], mod_gate=ms[:, 2 * D:3 * D], mod_stride=ms.stride(0), gate_prev=gprev,
                                      wT_qkv=blk.qkv.wT, wT_mlp=blk.proj_mlp.wT, wT_out=blk.proj_out.wT,
                                      A_cat_T=lo.A_cat_T if lo is not None else None, B_blk_T=lo.B_blk_T if lo is not None else None,
                                      norm_q=blk.norm_q, norm_k=blk.norm_k, cos_p=env.cos_p, sin_p=env.sin_p, key_bias=env.key_bias, dx=dx, dxg=dxg,
                                      g=mk(D) if dxg is None else None, dO=mk(D), dhpre=mk(4 * D), dn_mlp=mk(D), dqkv=mk(3 * D),
                                      U=mk(lo.K2) if lo is not None else None, dn=mk(D), dx_out=dx_out, dxg_out=dxg_out)
            if lo is not None and self.grad_sync is not None:
                self.grad_sync.ready(lo.flat_lo, lo.flat_hi)
            return dx_out, dxg_out, None, None
        if dm is not None:
            ops.colsum_prod(dx, dm[:, 2 * D:3 * D], b=sv.y, rows_per_batch=S)                        # d gate = sum_rows dOut * (un-gated branch output)
        g = dxg if dxg is not None else ops.scale_cols(dx, ms[:, 2 * D:3 * D], rpx)
        kw_o = kw_h = {}
        if lp is not None:         # proj_out's adapter: dx_in += (g sB) A, split over the two K segments [attn | mlp] of its input
            U_p = ops.gemm(g, lp.B_blk_T)
            kw_o = dict(a2=U_p, b2=lp.A_cat_T[:D], k2_real=lp.k2_real)
            kw_h = dict(a2=U_p, b2=lp.A_cat_T[D:], k2_real=lp.k2_real)
        dO = ops.gemm(g, blk.proj_out.wT[:D], **kw_o)
        dhpre = ops.gemm(g, blk.proj_out.wT[D:], epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre, **kw_h)
        if lp is not None:
            lp.grads([(sv.O, 0), (sv.hact, D)], sv.T_p, g, U_p, self.accumulate_lora_grads, self.grad_sync)
            del U_p
        if lm is not None:
            U_m = ops.gemm(dhpre, lm.B_blk_T)
            dn_mlp = ops.gemm(dhpre, blk.proj_mlp.wT, a2=U_m, b2=lm.A_cat_T, k2_real=lm.k2_real)
            lm.grads(sv.n, sv.T_m, dhpre, U_m, self.accumulate_lora_grads, self.grad_sync)
            del U_m
        else:
            dn_mlp = ops.gemm(dhpre, blk.proj_mlp.wT)
        del g, dhpre
        dqkv = torch.empty(B * S, 3 * D, dtype=BF16, device=dev)
        self._attn_rope_backward(sv, dO, dqkv, env, (blk.norm_q, blk.norm_k), (blk.norm_q, blk.norm_k), 0)
        del dO
        dn = self._lin_bwd(blk.qkv, dqkv, x=sv.n, T=sv.T, epilogue=EPI_ADD, aux_in=dn_mlp)
        if dm is not None:
            self._mod_grads(dn, sv.x, S, 0, 1, dm)
        d_txt = d_img = None
        if li > 0:
            gprev = msl(li - 1)[:, 2 * D:3 * D]
            dx, dxg = ops.ln_modulate_bwd(dn, sv.x, ms[:, D:2 * D], rpx, dres=dx, gate=gprev, want_gated=True)
        elif self.double:
            d_txt = torch.empty(B * St, D, dtype=BF16, device=dev); d_img = torch.empty(B * Si, D, dtype=BF16, device=dev)
            for b in range(B):
                for (r0, r1, dst) in ((b * S, b * S + St, d_txt[b * St:(b + 1) * St]), (b * S + St, (b + 1) * S, d_img[b * Si:(b + 1) * Si])):
                    ops.ln_modulate_bwd(dn[r0:r1], sv.x[r0:r1], ms[r0:r1, D:2 * D] if tokw else ms[b:b + 1, D:2 * D], 1 if tokw else r1 - r0, dres=dx[r0:r1], out=dst)
            dx = dxg = None
        else:
            dx, dxg = ops.ln_modulate_bwd(dn, sv.x, ms[:, D:2 * D], rpx, dres=dx)
        return dx, dxg, d_txt, d_img

    def _double_bwd(self, li: int, sv, d_img, d_txt, env):
        """backward of double block li (img / txt pairs as grouped launches).  Returns (d_img, d_txt) w.r.t. the block's inputs (None, None for
        block 0: the embedders are frozen, only its adapter gradients remain to compute)."""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        B, Si, St, S, mod, cos, sin = env.B, env.Si, env.St, env.S, env.mod, env.cos, env.sin
        blk = self.double[li]
        tokw = getattr(env, "tokenwise", False)       # per-token modulation rows on the image stream (see _double_fwd)
        mi = (env.mod_img if tokw else mod)[:, blk.mod_off:blk.mod_off + 6 * D]; mt = mod[:, blk.mod_off_c:blk.mod_off_c + 6 * D]
        rpi = 1 if tokw else Si
        ff = getattr(sv, "ff", None)              # '+ffs' adapters on this block's feed-forward Linears (host sequencing)
        dmi = dmt = None                          # 'ai-toolkit': the two streams' slices of d loss / d (modulation rows): chunks (shift, scale, gate)_msa, (shift, scale, gate)_mlp
        if self._dmod is not None:
            dmi, dmt = self._dmod[:, blk.mod_off:blk.mod_off + 6 * D], self._dmod[:, blk.mod_off_c:blk.mod_off_c + 6 * D]
        if (not tokw and ff is None and dmi is None and _block_abi_ok() and _BLOCK_ABI_ONLY in ("", "double", "bwd") and _FUSED_ROPE_BWD and li > 0 and sv.rrms is not None and not getattr(env, "routed", False) and sv.T_txt is None and sv.T_ao is None
                and Si % 256 == 0 and St % 256 == 0 and d_img.is_contiguous() and d_txt.is_contiguous()):
            # ONE C entry point (st355_block_flux_double_bwd): the launches of the host-side sequencing below, in its order, on its operands
            lq, lo_ = blk.qkv.lora, blk.to_out.lora
            mk = lambda r, c: torch.empty(r, c, dtype=BF16, device=dev)
            d_img_out, d_txt_out = mk(B * Si, D), mk(B * St, D)
            ops.block_flux_double_bwd(
                {"gA_qkv": list(lq.gA) if lq is not None else None, "gB_qkv": list(lq.gB) if lq is not None else None,
                 "gA_out": list(lo_.gA) if lo_ is not None else None, "gB_out": list(lo_.gB) if lo_ is not None else None},
                B=B, Si=Si, St=St, H=H, D=D, K2_qkv=lq.K2 if lq is not None else 0, k2r_qkv=lq.k2_real if lq is not None else 0,
                K2_out=lo_.K2 if lo_ is not None else 0, k2r_out=lo_.k2_real if lo_ is not None else 0,
                rank_qkv=lq.rank if lq is not None else 0, rpad_qkv=lq.r_pad if lq is not None else 0, rank_out=lo_.rank if lo_ is not None else 0,
                rpad_out=lo_.r_pad if lo_ is not None else 0, accumulate=1 if self.accumulate_lora_grads else 0, scale=env.scale,
                scale_qkv=lq.scale if lq is not None else 0.0, scale_out=lo_.scale if lo_ is not None else 0.0,
                img=sv.img, txt=sv.txt, n_img=sv.n_img, V=sv.V, rrms=sv.rrms, Q=sv.Q, K=sv.K, O=sv.O, lse2=sv.lse2, x1_img=sv.x1_img, x1_txt=sv.x1_txt,
                hpre_img=sv.hpre_img, hpre_txt=sv.hpre_txt, T_img=sv.T_img, T_o=sv.T_o, mod_img=mi, mod_txt=mt, mod_stride=mod.stride(0),
                wT_qkv=blk.qkv.wT, wT_add_qkv=blk.add_qkv.wT, wT_out=blk.to_out.wT, wT_add_out=blk.to_add_out.wT, wT_ff1=blk.ff1.wT, wT_ff2=blk.ff2.wT,
                wT_ffc1=blk.ffc1.wT, wT_ffc2=blk.ffc2.wT, At_qkv=lq.A_cat_T if lq is not None else None, Bbt_qkv=lq.B_blk_T if lq is not None else None,
                At_out=lo_.A_cat_T if lo_ is not None else None, Bbt_out=lo_.B_blk_T if lo_ is not None else None,
                norm_q=blk.norm_q, norm_k=blk.norm_k, norm_added_q=blk.norm_added_q, norm_added_k=blk.norm_added_k, cos_p=env.cos_p, sin_p=env.sin_p,
                key_bias=env.key_bias, d_img=d_img, d_txt=d_txt,
                g_img=mk(B * Si, D), g_txt=mk(B * St, D), dh_img=mk(B * Si, 4 * D), dh_txt=mk(B * St, 4 * D), dn2_img=mk(B * Si, D), dn2_txt=mk(B * St, D),
                dx1_img=mk(B * Si, D), dx1g_img=mk(B * Si, D), dx1_txt=mk(B * St, D), dx1g_txt=mk(B * St, D), dO=mk(B * S, D), dqkv=mk(B * S, 3 * D),
                U_qkv=mk(B * Si, lq.K2) if lq is not None else None, U_out=mk(B * Si, lo_.K2) if lo_ is not None else None,
                dn_img=mk(B * Si, D), dn_txt=mk(B * St, D), d_img_out=d_img_out, d_txt_out=d_txt_out)
            if self.grad_sync is not None:                     # the host-side order: to_out.0's adapters first, then to_q / to_k / to_v's
                for lg in (lo_, lq):
                    if lg is not None:
                        self.grad_sync.ready(lg.flat_lo, lg.flat_hi)
            return d_img_out, d_txt_out
        if dmi is not None:
            ops.colsum_prod(d_img, dmi[:, 5 * D:6 * D], b=sv.yf_i, rows_per_batch=Si)                  # d gate_mlp
            ops.colsum_prod(d_txt, dmt[:, 5 * D:6 * D], b=sv.yf_t, rows_per_batch=St)
        g_i = ops.scale_cols(d_img, mi[:, 5 * D:6 * D], rpi); g_t = ops.scale_cols(d_txt, mt[:, 5 * D:6 * D], St)
        if ff is None:
            dh_i, dh_t = ops.gemm_grouped([dict(a=g_i, w=blk.ff2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_img),
                                           dict(a=g_t, w=blk.ffc2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_txt)])
            dn2_i, dn2_t = ops.gemm_grouped([dict(a=dh_i, w=blk.ff1.wT), dict(a=dh_t, w=blk.ffc1.wT)])
        else:
            # dx = dy W + (dy sB) A as a K-extension of the dgrad launch (U = dy sB first); then the rank-space adapter gradients dB = s dy^T T, dA = U^T x
            back = lambda lg, dy: ops.gemm(dy, lg.B_blk_T) if lg is not None else None
            ext = lambda lg, U_: dict(a2=U_, b2=lg.A_cat_T, k2_real=lg.k2_real) if lg is not None else {}
            U_f2, U_c2 = back(blk.ff2.lora, g_i), back(blk.ffc2.lora, g_t)
            dh_i, dh_t = ops.gemm_grouped([dict(a=g_i, w=blk.ff2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_img, **ext(blk.ff2.lora, U_f2)),
                                           dict(a=g_t, w=blk.ffc2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_txt, **ext(blk.ffc2.lora, U_c2))])
            for (lg, x_in, T_, dy, U_) in ((blk.ff2.lora, ff.h_i, ff.T_f2, g_i, U_f2), (blk.ffc2.lora, ff.h_t, ff.T_c2, g_t, U_c2)):
                if lg is not None:
                    lg.grads(x_in, T_, dy, U_, self.accumulate_lora_grads, self.grad_sync)
            U_f1, U_c1 = back(blk.ff1.lora, dh_i), back(blk.ffc1.lora, dh_t)
            dn2_i, dn2_t = ops.gemm_grouped([dict(a=dh_i, w=blk.ff1.wT, **ext(blk.ff1.lora, U_f1)), dict(a=dh_t, w=blk.ffc1.wT, **ext(blk.ffc1.lora, U_c1))])
            for (lg, x_in, T_, dy, U_) in ((blk.ff1.lora, ff.n2_i, ff.T_f1, dh_i, U_f1), (blk.ffc1.lora, ff.n2_t, ff.T_c1, dh_t, U_c1)):
                if lg is not None:
                    lg.grads(x_in, T_, dy, U_, self.accumulate_lora_grads, self.grad_sync)
            del U_f2, U_c2, U_f1, U_c1
        del g_i, g_t, dh_i, dh_t
        if dmi is not None:
            self._mod_grads(dn2_i, sv.x1_img, Si, 3, 4, dmi); self._mod_grads(dn2_t, sv.x1_txt, St, 3, 4, dmt)
        dx1_i, dx1g_i = ops.ln_modulate_bwd(dn2_i, sv.x1_img, mi[:, 4 * D:5 * D], rpi, dres=d_img, gate=mi[:, 2 * D:3 * D], want_gated=True)
        dx1_t, dx1g_t = ops.ln_modulate_bwd(dn2_t, sv.x1_txt, mt[:, 4 * D:5 * D], St, dres=d_txt, gate=mt[:, 2 * D:3 * D], want_gated=True)
        del dn2_i, dn2_t
        if dmi is not None:
            ops.colsum_prod(dx1_i, dmi[:, 2 * D:3 * D], b=sv.ya_i, rows_per_batch=Si)                  # d gate_msa
            ops.colsum_prod(dx1_t, dmt[:, 2 * D:3 * D], b=sv.ya_t, rows_per_batch=St)
        # attention output projections: dO rows of both streams (+ adapter grads)
        dO = torch.empty(B * S, D, dtype=BF16, device=dev)
        U_i = ops.gemm(dx1g_i, blk.to_out.lora.B_blk_T) if blk.to_out.lora is not None else None
        U_t = ops.gemm(dx1g_t, blk.to_add_out.lora.B_blk_T) if blk.to_add_out.lora is not None else None
        kw_i = dict(a2=U_i, b2=blk.to_out.lora.A_cat_T, k2_real=blk.to_out.lora.k2_real) if U_i is not None else {}
        kw_t = dict(a2=U_t, b2=blk.to_add_out.lora.A_cat_T, k2_real=blk.to_add_out.lora.k2_real) if U_t is not None else {}
        ops.gemm_grouped(self._problems(env, Si, dict(a=dx1g_i, w=blk.to_out.wT, out=self._rows_of(dO, St, Si, env), **kw_i))
                         + self._problems(env, St, dict(a=dx1g_t, w=blk.to_add_out.wT, out=self._rows_of(dO, 0, St, env), **kw_t)))
        for (lin, U, T_, dxg, lo, rows) in ((blk.to_out, U_i, sv.T_o, dx1g_i, St, Si), (blk.to_add_out, U_t, sv.T_ao, dx1g_t, 0, St)):
            if lin.lora is not None:       # dA = U^T O over this stream's rows of the joint attention output, read in place
                lin.lora.grads(self._compact(self._rows_of(sv.O, lo, rows, env), env, rows), T_, dxg, U, self.accumulate_lora_grads, self.grad_sync)
        del dx1g_i, dx1g_t, U_i, U_t
        dqkv = torch.empty(B * S, 3 * D, dtype=BF16, device=dev)
        self._attn_rope_backward(sv, dO, dqkv, env, (blk.norm_added_q, blk.norm_added_k), (blk.norm_q, blk.norm_k), St)
        del dO
        last = li == 0 and self.l_x.lora is None and dmi is None      # (an adapter on x_embedder needs the image stream's input gradient of block 0; the modulation adapters d n of both streams)
        # the two streams' rows of the joint dqkv, in place (the reference's autograd splits the concatenated gradient the same way)
        dq_i, dq_t = self._rows_of(dqkv, St, Si, env), self._rows_of(dqkv, 0, St, env)
        streams = [("img", blk.qkv, dq_i, sv.n_img, sv.T_img, Si), ("txt", blk.add_qkv, dq_t, sv.n_txt, sv.T_txt, St)]
        if last:
            streams = [s_ for s_ in streams if s_[1].lora is not None]   # frozen embedders: only adapter grads remain to compute
        probs, Us, dns = [], {}, []
        for (name, lin, dq, n_in, T_, rows) in streams:
            kw = {}
            if lin.lora is not None:
                Us[name] = torch.empty(B * rows, lin.lora.B_blk_T.shape[0], dtype=BF16, device=dev)
                for pr in self._problems(env, rows, dict(a=dq, w=lin.lora.B_blk_T, out=Us[name])):
                    ops.gemm(pr.pop("a"), pr.pop("w"), **pr)
                kw = dict(a2=Us[name], b2=lin.lora.A_cat_T, k2_real=lin.lora.k2_real)
            if not last:
                dns.append(torch.empty(B * rows, D, dtype=BF16, device=dev))
                probs += self._problems(env, rows, dict(a=dq, w=lin.wT, out=dns[-1], **kw))
        if probs:
            ops.gemm_grouped(probs)
        for (name, lin, dq, n_in, T_, rows) in streams:
            if lin.lora is not None:
                lin.lora.grads(n_in, T_, self._compact(dq, env, rows), Us[name], self.accumulate_lora_grads, self.grad_sync)
        if last:
            return None, None
        if dmi is not None:
            self._mod_grads(dns[0], sv.img, Si, 0, 1, dmi); self._mod_grads(dns[1], sv.txt, St, 0, 1, dmt)
        d_img, _ = ops.ln_modulate_bwd(dns[0], sv.img, mi[:, D:2 * D], rpi, dres=dx1_i)
        d_txt, _ = ops.ln_modulate_bwd(dns[1], sv.txt, mt[:, D:2 * D], St, dres=dx1_t)
        return d_img, d_txt

    def _engine_backward(self, ctx, dout):
        """hand-written backward for frozen-base (LoRA) training: dX chain + rank-space adapter gradients.  Checkpointed segments are re-run from
        their kept input first (same kernels, same order: the recomputed activations are bit-identical to the ones a plain forward keeps)."""
        if not self._prepared:
            raise RuntimeError("call prepare_for_training() after loading weights (builds the K-major dgrad operands)")
        D = self.D
        env = ctx.env
        B, Si, St, S, mod = env.B, env.Si, env.St, env.S, env.mod
        dev = self.device_
        dout = dout.reshape(B * Si, -1).to(BF16).contiguous()
        mo = mod[:, self.mod_off_out:self.mod_off_out + 2 * D]
        dn = self._lin_bwd(self.l_out, dout, x=getattr(ctx, "n_out", None), T=getattr(ctx, "T_out", None))
        self._dmod = torch.zeros(B, self.mod_total, dtype=F32, device=dev) if self.mod_lora is not None else None
        dx = torch.zeros(B * S, D, dtype=BF16, device=dev)      # the txt rows of the last single block get no gradient
        for b in range(B):          # written straight into the image rows of the joint gradient
            if getattr(env, "tokenwise", False):          # norm_out's (scale, shift) rows are per image token (flux/transformer.py:1505 takes temb_img)
                ops.ln_modulate_bwd(dn[b * Si:(b + 1) * Si], ctx.x_final[b * S + St:(b + 1) * S], env.mod_img[b * Si:(b + 1) * Si, self.mod_off_out:self.mod_off_out + D], 1,
                                    out=dx[b * S + St:(b + 1) * S])
                continue
            ops.ln_modulate_bwd(dn[b * Si:(b + 1) * Si], ctx.x_final[b * S + St:(b + 1) * S], mo[b:b + 1, :D], Si, out=dx[b * S + St:(b + 1) * S])
        del dn
        ctx.x_final = None
        # ---- single blocks, reversed ----
        # TREAD: the backward enters a route at its END block (the routed blocks see only the kept tokens' gradient rows; the skipped tokens' gradient stays in
        # d_full) and leaves it at its START block (the kept rows go back into d_full: the gradient of the full stream at the route's start)
        nd = len(self.double)
        stop = getattr(self, "_bwd_stop", 0)          # global index of the first block with an adapter (add_lora_adapter)
        dxg = d_txt = d_img = d_full = None
        keep_of = lambda info: info.keep_i32()
        for (s0, n, ck) in reversed(ctx.segs_s):
            if ck:
                # a segment that straddles the stop block: blocks below it are re-run (their outputs feed the ones above) but keep nothing — the backward returns at `stop`
                xr = ctx.ck_s.pop(s0)
                for bi in range(s0, s0 + n):
                    xr, sv = self._single_fwd(bi, xr, ctx.env_s[bi], nd + bi >= stop)
                    if nd + bi >= stop:
                        ctx.sgl[bi] = sv
                del xr
            for li in range(s0 + n - 1, s0 - 1, -1):
                g, e = nd + li, ctx.env_s[li]
                if g in ctx.route_end:
                    dxv = dx.view(B, S, D)
                    d_full = dxv[:, St:].contiguous()
                    dx = torch.cat([dxv[:, :St], ops.gather_rows(d_full, keep_of(ctx.route_end[g]))], dim=1).reshape(-1, D)
                    dxg = None
                dx, dxg, d_txt, d_img = self._single_bwd(li, ctx.sgl.pop(li), dx, dxg, e)
                if g == stop and g > 0:
                    ctx.sgl.clear(); ctx.dbl.clear(); ctx.ck_s.clear(); ctx.ck_d.clear()
                    return None
                if g in ctx.route_start:
                    if dx is None:                               # single block 0 under double blocks: its input gradient came back stream-major
                        ops.scatter_rows(d_img.view(B, e.Si, D), keep_of(ctx.route_start[g]), d_full)
                        d_img, d_full = d_full.view(-1, D), None
                    else:
                        dxv = dx.view(B, e.S, D)
                        ops.scatter_rows(dxv[:, St:].contiguous(), keep_of(ctx.route_start[g]), d_full)
                        dx = torch.cat([dxv[:, :St], d_full], dim=1).reshape(B * S, D)
                        dxg, d_full = None, None
        # ---- split the joint gradient (only when there was no single block to do it) ----
        if not self.single:
            d_txt = dx.view(B, S, D)[:, :St].reshape(B * St, D); d_img = dx.view(B, S, D)[:, St:].reshape(B * Si, D)
        # ---- double blocks, reversed ----
        for (s0, n, ck) in reversed(ctx.segs_d):
            if ck:
                ir, tr = ctx.ck_d.pop(s0)
                for bi in range(s0, s0 + n):
                    ir, tr, _, sv = self._double_fwd(bi, ir, tr, ctx.env_d[bi], bi >= stop)
                    if bi >= stop:
                        ctx.dbl[bi] = sv
                del ir, tr
            for li in range(s0 + n - 1, s0 - 1, -1):
                e = ctx.env_d[li]
                if li in ctx.route_end:
                    d_full = d_img.view(B, Si, D)
                    d_img = ops.gather_rows(d_full, keep_of(ctx.route_end[li])).view(-1, D)
                d_img, d_txt = self._double_bwd(li, ctx.dbl.pop(li), d_img, d_txt, e)
                if li == stop and li > 0:
                    ctx.dbl.clear(); ctx.ck_d.clear()
                    return None
                if li in ctx.route_start and d_img is not None:
                    ops.scatter_rows(d_img.view(B, e.Si, D), keep_of(ctx.route_start[li]), d_full)
                    d_img, d_full = d_full.reshape(-1, D), None
        lx = self.l_x.lora
        if lx is not None:          # 'all+ffs+embedder': dy of x_embedder = the image stream's gradient at block 0's input; its own input (the packed latents) needs none
            if d_img is None:       # no double block: the image rows of the joint gradient at single block 0's input
                d_img = dx.view(B, S, D)[:, St:].reshape(B * Si, D)
            U_x = ops.gemm(d_img, lx.B_blk_T)
            lx.grads(ctx.x2d, ctx.T_x, d_img, U_x, self.accumulate_lora_grads, self.grad_sync)
        ml = self.mod_lora
        if ml is not None:          # 'ai-toolkit': the modulation Linears' adapters — dy = the accumulated modulation-row gradient [B, mod_total], x = silu(temb) [B, D]
            dmod = self._dmod.to(BF16)
            self._dmod = None
            U_m = ops.gemm(dmod, ml.B_blk_T)
            ml.grads(ctx.st, ctx.T_mod, dmod, U_m, self.accumulate_lora_grads, self.grad_sync)
        return None

    # ------------------------------------------------------------------------------------------------
    # full-rank training (`model_type == "full"`: the reference's multi-GPU Flux datapoint trains the whole transformer, documentation/DISTRIBUTED.md:291-298):
    # every weight, bias, q / k RMSNorm weight and modulation row trains.  bf16 parameters and bf16 gradients in two arenas of one layout.
    # ------------------------------------------------------------------------------------------------
    def _all_linears(self):
        ls = [self.l_x, self.l_ctx, self.l_t1, self.l_t2, self.l_p1, self.l_p2, self.l_out]
        if self.config.guidance_embeds:
            ls += [self.l_g1, self.l_g2]
        for blk in self.double:
            ls += [blk.qkv, blk.add_qkv, blk.to_out, blk.to_add_out, blk.ff1, blk.ff2, blk.ffc1, blk.ffc2]
        for blk in self.single:
            ls += [blk.qkv, blk.proj_mlp, blk.proj_out]
        return ls

    def enable_full_finetune(self):
        """gradient arena with the weight arena's layout; every base parameter becomes trainable (one fused optimizer launch, contiguous exchange slices)"""
        if self.lora_groups:
            raise RuntimeError("full-rank training and LoRA adapters are exclusive")
        self.full = True
        self.grad_arena = torch.zeros_like(self.arena)
        base = self.arena.data_ptr()

        def gview(t):
            off = (t.data_ptr() - base) // 2
            return self.grad_arena[off:off + t.numel()].view(t.shape)

        for l in self._all_linears():
            l.gw, l.gb = gview(l.w), gview(l.b)
        for blk in self.double + self.single:
            for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                w = getattr(blk, nm, None)
                if w is not None:
                    setattr(blk, "g_" + nm, gview(w))
        self.g_mod_w, self.g_mod_b = gview(self.mod_w), gview(self.mod_b)
        ps = sorted([p for n, p in self.named_parameters() if ".lora_" not in n], key=lambda p: p.data_ptr())
        for p in ps:
            p.requires_grad_(True)
        self._full_params = ps
        self._full_offsets = [((p.data_ptr() - base) // 2, p.numel()) for p in ps]
        self.prepare_for_training()
        return ps

    def _refresh_transposed(self):
        """W^T follows the weights (2 B read + 2 B write per parameter per step: ~12 ms for Flux.1-dev's 12 B parameters)"""
        for l in self._all_linears():
            if getattr(l, "wT", None) is not None:
                ops.transpose(l.w, out=l.wT)

    def diffusers_state_dict(self) -> Dict[str, torch.Tensor]:
        """{diffusers checkpoint key: tensor} of the base parameters (the parameter names ARE the checkpoint keys): what `save_pretrained` writes"""
        return {k: v.detach() for k, v in self.named_parameters() if ".lora_" not in k}

    def load_diffusers_state(self, state: Dict[str, torch.Tensor]):
        self.load_flat_state(state)

    def _single_bwd_full(self, li: int, sv, dx, env, fb):
        """backward of single block li with every parameter gradient; returns the gradient of the block's input (joint [txt || img] rows)"""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        B, S, mod = env.B, env.S, env.mod
        blk = self.single[li]
        ms = mod[:, blk.mod_off:blk.mod_off + 3 * D]; dms = fb.dmod[:, blk.mod_off:blk.mod_off + 3 * D]
        ops.colsum_prod(dx, dms[:, 2 * D:3 * D], b=sv.y, rows_per_batch=S)                         # d gate
        g = ops.scale_cols(dx, ms[:, 2 * D:3 * D], S)
        # proj_out reads [attention output | MLP activation] as two K segments: its weight gradient is written as the two column blocks
        gp = fb.P64(g)
        ops.gemm_tn(gp, fb.P64(sv.O), out=blk.proj_out.gw[:, :D])
        ops.gemm_tn(gp, fb.P64(sv.hact), out=blk.proj_out.gw[:, D:])
        fb.bgrad(blk.proj_out, g)
        dO = ops.gemm(g, blk.proj_out.wT[:D])
        dhpre = ops.gemm(g, blk.proj_out.wT[D:], epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre)
        del g, gp
        fb.wgrad(blk.proj_mlp, dhpre, sv.n)
        dn_mlp = ops.gemm(dhpre, blk.proj_mlp.wT)
        del dhpre
        dqkv = torch.empty(B * S, 3 * D, dtype=BF16, device=dev)
        dQ, dK = self._attn_backward(sv, dO, dqkv, env)
        ops.qk_norm_rope_bwd_wgrad(dQ, dK, sv.qkv, blk.norm_q, blk.norm_k, env.cos, env.sin, dqkv, B, H, hd, S, 0, S, blk.g_norm_q, blk.g_norm_k)
        del dQ, dK, dO
        fb.wgrad(blk.qkv, dqkv, sv.n)
        dn = ops.gemm(dqkv, blk.qkv.wT, epilogue=EPI_ADD, aux_in=dn_mlp)
        fb.mod_grads(dn, sv.x, S, 0, 1, dms)
        dx_in, _ = ops.ln_modulate_bwd(dn, sv.x, ms[:, D:2 * D], S, dres=dx)
        return dx_in

    def _double_bwd_full(self, li: int, sv, d_img, d_txt, env, fb):
        """backward of double block li with every parameter gradient; returns (d_img, d_txt) w.r.t. the block's inputs"""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        B, Si, St, S, mod = env.B, env.Si, env.St, env.S, env.mod
        blk = self.double[li]
        mi = mod[:, blk.mod_off:blk.mod_off + 6 * D]; mt = mod[:, blk.mod_off_c:blk.mod_off_c + 6 * D]
        dmi = fb.dmod[:, blk.mod_off:blk.mod_off + 6 * D]; dmt = fb.dmod[:, blk.mod_off_c:blk.mod_off_c + 6 * D]
        rows = lambda t, lo, n: t[lo:lo + n] if B == 1 else t.view(B, S, -1)[:, lo:lo + n].reshape(B * n, -1)
        # ---- MLP branches ----
        ops.colsum_prod(d_img, dmi[:, 5 * D:6 * D], b=sv.yf_i, rows_per_batch=Si)                  # d gate_mlp
        ops.colsum_prod(d_txt, dmt[:, 5 * D:6 * D], b=sv.yf_t, rows_per_batch=St)
        g_i = ops.scale_cols(d_img, mi[:, 5 * D:6 * D], Si); g_t = ops.scale_cols(d_txt, mt[:, 5 * D:6 * D], St)
        fb.wgrad(blk.ff2, g_i, sv.h_i); fb.wgrad(blk.ffc2, g_t, sv.h_t)
        dh_i, dh_t = ops.gemm_grouped([dict(a=g_i, w=blk.ff2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_img),
                                       dict(a=g_t, w=blk.ffc2.wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.hpre_txt)])
        fb.wgrad(blk.ff1, dh_i, sv.n2_i); fb.wgrad(blk.ffc1, dh_t, sv.n2_t)
        dn2_i, dn2_t = ops.gemm_grouped([dict(a=dh_i, w=blk.ff1.wT), dict(a=dh_t, w=blk.ffc1.wT)])
        del g_i, g_t, dh_i, dh_t
        fb.mod_grads(dn2_i, sv.x1_img, Si, 3, 4, dmi); fb.mod_grads(dn2_t, sv.x1_txt, St, 3, 4, dmt)
        dx1_i, dx1g_i = ops.ln_modulate_bwd(dn2_i, sv.x1_img, mi[:, 4 * D:5 * D], Si, dres=d_img, gate=mi[:, 2 * D:3 * D], want_gated=True)
        dx1_t, dx1g_t = ops.ln_modulate_bwd(dn2_t, sv.x1_txt, mt[:, 4 * D:5 * D], St, dres=d_txt, gate=mt[:, 2 * D:3 * D], want_gated=True)
        del dn2_i, dn2_t
        ops.colsum_prod(dx1_i, dmi[:, 2 * D:3 * D], b=sv.ya_i, rows_per_batch=Si)                  # d gate_msa
        ops.colsum_prod(dx1_t, dmt[:, 2 * D:3 * D], b=sv.ya_t, rows_per_batch=St)
        # ---- attention output projections (joint order [txt || img]) ----
        fb.wgrad(blk.to_out, dx1g_i, rows(sv.O, St, Si)); fb.wgrad(blk.to_add_out, dx1g_t, rows(sv.O, 0, St))
        dO = torch.empty(B * S, D, dtype=BF16, device=dev)
        ops.gemm_grouped(self._problems(env, Si, dict(a=dx1g_i, w=blk.to_out.wT, out=self._rows_of(dO, St, Si, env)))
                         + self._problems(env, St, dict(a=dx1g_t, w=blk.to_add_out.wT, out=self._rows_of(dO, 0, St, env))))
        del dx1g_i, dx1g_t
        # ---- attention, RoPE / RMSNorm (+ norm weight gradients) ----
        dqkv = torch.empty(B * S, 3 * D, dtype=BF16, device=dev)
        dQ, dK = self._attn_backward(sv, dO, dqkv, env)
        ops.qk_norm_rope_bwd_wgrad(dQ, dK, sv.qkv, blk.norm_added_q, blk.norm_added_k, env.cos, env.sin, dqkv, B, H, hd, St, 0, S, blk.g_norm_added_q, blk.g_norm_added_k)
        ops.qk_norm_rope_bwd_wgrad(dQ, dK, sv.qkv, blk.norm_q, blk.norm_k, env.cos, env.sin, dqkv, B, H, hd, Si, St, S, blk.g_norm_q, blk.g_norm_k)
        del dQ, dK, dO
        dq_i, dq_t = rows(dqkv, St, Si), rows(dqkv, 0, St)
        fb.wgrad(blk.qkv, dq_i, sv.n_img); fb.wgrad(blk.add_qkv, dq_t, sv.n_txt)
        dn_i, dn_t = ops.gemm_grouped([dict(a=dq_i, w=blk.qkv.wT), dict(a=dq_t, w=blk.add_qkv.wT)])
        fb.mod_grads(dn_i, sv.img, Si, 0, 1, dmi); fb.mod_grads(dn_t, sv.txt, St, 0, 1, dmt)
        d_img_in, _ = ops.ln_modulate_bwd(dn_i, sv.img, mi[:, D:2 * D], Si, dres=dx1_i)
        d_txt_in, _ = ops.ln_modulate_bwd(dn_t, sv.txt, mt[:, D:2 * D], St, dres=dx1_t)
        return d_img_in, d_txt_in

    def _engine_backward_full(self, ctx, dout):
        """hand-written backward of full-rank training: the dX chain of `_engine_backward` plus a TN weight-gradient GEMM and a bias column sum per Linear,
        token-axis reductions for the modulation rows / gates, the q / k RMSNorm weight gradients, and the embedders.  Checkpointed segments are re-run from
        their kept input first (same kernels, same order, bit-identical activations)."""
        D, H, hd, dev = self.D, self.H, self.hd, self.device_
        env = ctx.env
        B, Si, St, S, mod = env.B, env.Si, env.St, env.S, env.mod
        self._refresh_transposed()
        fb = SimpleNamespace(dmod=torch.zeros(B, self.mod_total, dtype=F32, device=dev))        # d loss / d (modulation linear output)
        tmp_b = {}

        def P64(t):
            """zero-padded copy with a multiple of 64 rows (the TN GEMM's contraction granule); no copy when already aligned"""
            r = t.shape[0]
            if r % 64 == 0 and t.is_contiguous():
                return t
            o = torch.zeros((r + 63) // 64 * 64, t.shape[1], dtype=BF16, device=dev)
            o[:r] = t
            return o

        def bgrad(lin, dy):
            N = dy.shape[1]
            t = tmp_b.get(N)
            if t is None:
                t = tmp_b[N] = torch.empty(1, N, dtype=F32, device=dev)
            ops.colsum_prod(dy, t)
            lin.gb.copy_(t[0])

        def wgrad(lin, dy, x):
            """dW = dY^T X ; db = colsum(dY)   (into the gradient arena views of `lin`)"""
            ops.gemm_tn(P64(dy), P64(x), out=lin.gw)
            bgrad(lin, dy)

        def mod_grads(dn, x_in, rows, k_shift, k_scale, dm, xhat=None):
            """d shift = sum_t dY, d scale = sum_t dY * LN(x) of one AdaLN instance (chunk indices k_* inside its slice dm of the modulation gradient);
            x_in = the LayerNorm's input (or xhat = LN(x) when the caller already has it)"""
            ops.colsum_prod(dn, dm[:, k_shift * D:(k_shift + 1) * D], rows_per_batch=rows)
            ops.colsum_prod(dn, dm[:, k_scale * D:(k_scale + 1) * D], b=ops.layer_norm_xhat(x_in) if xhat is None else xhat, rows_per_batch=rows)

        fb.P64, fb.bgrad, fb.wgrad, fb.mod_grads = P64, bgrad, wgrad, mod_grads
        sync = self.grad_sync
        # ---- output head (AdaLayerNormContinuous: chunk order (scale, shift)) ----
        dout = dout.reshape(B * Si, -1).to(BF16).contiguous()
        mo = mod[:, self.mod_off_out:self.mod_off_out + 2 * D]
        dmo = fb.dmod[:, self.mod_off_out:self.mod_off_out + 2 * D]
        wgrad(self.l_out, dout, ctx.n_out)
        dn = ops.gemm(dout, self.l_out.wT)
        xhat = torch.empty(B * Si, D, dtype=BF16, device=dev)
        for b in range(B):          # LN of the image rows of the joint sequence (strided per-sample views)
            ops.layer_norm_xhat(ctx.x_final[b * S + St:(b + 1) * S], out=xhat[b * Si:(b + 1) * Si])
        mod_grads(dn, None, Si, 1, 0, dmo, xhat=xhat)
        del xhat
        dx = torch.zeros(B * S, D, dtype=BF16, device=dev)      # the txt rows of the last single block get no gradient
        for b in range(B):
            ops.ln_modulate_bwd(dn[b * Si:(b + 1) * Si], ctx.x_final[b * S + St:(b + 1) * S], mo[b:b + 1, :D], Si, out=dx[b * S + St:(b + 1) * S])
        del dn
        ctx.x_final = ctx.n_out = None
        if sync is not None:
            sync.ready(self._head_arena_lo, self.grad_arena.numel())        # proj_out gradients are final
        # The fused modulation matrix (every block's adaLN Linear as rows of ONE [mod_total, D] matrix: 3.2 B of Flux.1-dev's 11.9 B parameters, 6.5 GB of gradient)
        # gets its gradient rows block by block (r6): dW_mod[r0:r1] = dmod[:, r0:r1]^T silu(temb) as soon as the block that owns rows [r0, r1) has run, so that
        # the exchange takes them behind the backward instead of as one exposed region after it.  Same arithmetic per row (one 64-deep contraction over the
        # zero-padded batch): bit-equal to the one-product form.
        Bp = (B + 63) // 64 * 64
        st_p = torch.zeros(Bp, D, dtype=BF16, device=dev); st_p[:B] = ctx.emb.st
        mw_lo = (self.mod_w.data_ptr() - self.arena.data_ptr()) // 2
        mw_hi = mw_lo + self.mod_total * D
        front_hi = self.double[0].arena_lo if self.double else (self.single[0].arena_lo if self.single else self._head_arena_lo)
        mod_in_front = 0 <= mw_lo and mw_hi <= front_hi
        mod_rows_lo = [self.mod_total]                    # rows [mod_rows_lo, mod_total) of dW_mod are written (and handed over)

        def mod_rows_grad(r0):
            r1 = mod_rows_lo[0]
            if r1 <= r0:
                return
            dp = torch.zeros(Bp, r1 - r0, dtype=BF16, device=dev); dp[:B] = fb.dmod[:, r0:r1]
            ops.gemm_tn(dp, st_p, out=self.g_mod_w[r0:r1])
            mod_rows_lo[0] = r0
            if sync is not None and mod_in_front:
                sync.ready(mw_lo + r0 * D, mw_lo + r1 * D)

        mod_rows_grad(self.mod_off_out)                   # norm_out's (scale, shift) rows: final since mod_grads above
        # ---- single blocks, reversed ----
        for (s0, n, ck) in reversed(ctx.segs_s):
            if ck:
                xr = ctx.ck_s.pop(s0)
                for bi in range(s0, s0 + n):
                    xr, ctx.sgl[bi] = self._single_fwd(bi, xr, ctx.env_s[bi], True)
                del xr
            for li in range(s0 + n - 1, s0 - 1, -1):
                dx = self._single_bwd_full(li, ctx.sgl.pop(li), dx, ctx.env_s[li], fb)
                if sync is not None:
                    sync.ready(self.single[li].arena_lo, self.single[li].arena_hi)
                mod_rows_grad(self.single[li].mod_off)
        # ---- split the joint gradient [txt || img] ----
        dxv = dx.view(B, S, D)
        d_txt, d_img = dxv[:, :St].reshape(B * St, D), dxv[:, St:].reshape(B * Si, D)
        del dx, dxv
        # ---- double blocks, reversed ----
        for (s0, n, ck) in reversed(ctx.segs_d):
            if ck:
                ir, tr = ctx.ck_d.pop(s0)
                for bi in range(s0, s0 + n):
                    ir, tr, _, ctx.dbl[bi] = self._double_fwd(bi, ir, tr, ctx.env_d[bi], True)
                del ir, tr
            for li in range(s0 + n - 1, s0 - 1, -1):
                d_img, d_txt = self._double_bwd_full(li, ctx.dbl.pop(li), d_img, d_txt, ctx.env_d[li], fb)
                if sync is not None:
                    sync.ready(self.double[li].arena_lo, self.double[li].arena_hi)
                mod_rows_grad(self.double[li].mod_off)
        # ---- embedders ----
        em = ctx.emb
        wgrad(self.l_x, d_img, em.x2d)
        wgrad(self.l_ctx, d_txt, em.enc2d)
        # modulation linear: mod = silu(temb) W_mod^T + b
        dmod_p = torch.zeros(Bp, self.mod_total, dtype=BF16, device=dev); dmod_p[:B] = fb.dmod
        mod_rows_grad(0)                                                    # whatever rows are left (none when the model has blocks: the last block handed over row 0)
        tb = torch.empty(1, self.mod_total, dtype=F32, device=dev)
        ops.colsum_prod(dmod_p, tb)
        self.g_mod_b.copy_(tb[0])
        # d silu(temb) = dmod @ W_mod -> [B, D], as (W_mod^T dmod^T)^T with the TN GEMM.  The contraction runs over the mod_total rows of W_mod (1.06 M for
        # Flux.1-dev): walked in row blocks that stay inside the TN GEMM's 2 GiB operand window, accumulated into one [D, B8] output
        B8 = 8 * ((B + 7) // 8)
        dmod_t = ops.transpose(dmod_p[:B8])                                   # [mod_total, B8]
        blk_rows = max(64, (getattr(self, "_tn_window_bytes", (1 << 31) - 1) // (2 * max(D, B8))) // 64 * 64)        # (attribute: tests shrink the window)
        acc = torch.empty(D, B8, dtype=BF16, device=dev)
        for i, r0 in enumerate(range(0, self.mod_total, blk_rows)):
            r1 = min(self.mod_total, r0 + blk_rows)
            ops.gemm_tn(self.mod_w[r0:r1], dmod_t[r0:r1], out=acc, accumulate=i > 0)
        dst = ops.transpose(acc)[:B].contiguous()
        dtemb = ops.silu_bwd(em.temb, dst)

        def mlp_bwd(l1, l2, x_in, pre1, act1, dy):
            """TimestepEmbedding / guidance / pooled-text projection: y = l2(silu(l1(x)))"""
            pad = lambda t: torch.cat([t, torch.zeros(Bp - B, t.shape[1], dtype=BF16, device=dev)], dim=0)
            dyp = pad(dy)
            ops.gemm_tn(dyp, pad(act1), out=l2.gw)
            bgrad(l2, dyp)
            d1p = pad(ops.silu_bwd(pre1, ops.gemm(dy, l2.wT)))
            ops.gemm_tn(d1p, pad(x_in), out=l1.gw)
            bgrad(l1, d1p)

        mlp_bwd(self.l_t1, self.l_t2, em.tproj, em.t1, em.st1, dtemb)
        if self.config.guidance_embeds:
            mlp_bwd(self.l_g1, self.l_g2, em.gproj, em.g1, em.sg1, dtemb)
        mlp_bwd(self.l_p1, self.l_p2, em.pooled, em.p1, em.sp1, dtemb)
        if sync is not None:
            if mod_in_front:                                                   # embedders + the modulation bias: what lies around the modulation matrix's rows
                sync.ready(mw_hi, front_hi)
                sync.ready(0, mw_lo)
            else:
                sync.ready(0, front_hi)                                        # embedders + the modulation matrix
        return None

    # ------------------------------------------------------------------------------------------------
    # public forward (reference signature: flux/transformer.py:940-960)
    # ------------------------------------------------------------------------------------------------
    def set_router(self, router, routes):
        """flux/transformer.py:829-831: TREAD router + [{selection_ratio, start_layer_idx, end_layer_idx}] (training/tread.py)"""
        self._tread_router, self._tread_routes = router, routes

    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, img_ids=None, txt_ids=None,
                guidance=None, joint_attention_kwargs=None, return_dict: bool = True, attention_mask=None, force_keep_mask=None, rope_layout_key=None, **unsupported):
        self._rope_layout_key = rope_layout_key            # hashable name of the id layout (see _rope); None = compute the tables from the ids
        self._force_keep_mask = force_keep_mask            # TREAD: tokens that may never be routed away (flux/transformer.py:958, 1216-1220)
        for k, v in unsupported.items():
            if v is not None and v is not False:
                raise NotImplementedError(f"FluxTransformer2DModel(st355): argument {k!r} is not supported on the HIP path")
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        if timestep.ndim not in (1, 2):
            raise ValueError(f"timestep: expected [B] or tokenwise [B, S_img], got {tuple(timestep.shape)}")
        key_bias = None
        if attention_mask is not None:
            # flux_attention_masked_training (flux/model.py:813-823): the text mask [B, S_txt] is expanded with ones over the image tokens
            # (expand_flux_attention_mask, flux/transformer.py:227-242) and handed to SDPA as `(mask > 0).bool().to(hidden_states.dtype)`
            # (:170-173) — a FLOAT mask, which SDPA ADDS to the scores: valid keys get +1.0, padded text keys +0.0.  That additive form is
            # what the reference trains with, so it is what the attention kernels' per-key bias reproduces here.
            am = attention_mask
            if am.dim() == 3 and am.size(1) == 1:
                am = am.squeeze(1)
            B_, S_tot = hidden_states.shape[0], hidden_states.shape[1] + encoder_hidden_states.shape[1]
            key_bias = torch.ones(B_, S_tot, dtype=F32, device=self.device_)
            key_bias[:, :am.shape[1]] = (am.to(self.device_) > 0).to(F32)
        need_grad = torch.is_grad_enabled() and (len(self._lora_params) > 0 or self.full)
        if need_grad and not self._prepared:
            # the K-major dgrad operands follow the weights: load_flat_state / init_synthetic / the replica start-state broadcast
            # (training.grad_sync.sync_module_states) mark them stale, the next training forward rebuilds them
            self.prepare_for_training()
        if need_grad and self.full:
            out = _FluxFullFn.apply(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, guidance, img_ids, txt_ids, key_bias,
                                    *self._full_params)
        elif need_grad:
            out = _FluxFn.apply(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, guidance, img_ids, txt_ids, key_bias,
                                *self._lora_params)
        else:
            with torch.no_grad():
                out, _ = self._engine_forward(hidden_states.to(BF16), encoder_hidden_states.to(BF16), pooled_projections, timestep,
                                              guidance, img_ids, txt_ids, save=False, key_bias=key_bias)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)


class _FluxFn(torch.autograd.Function):
    """one autograd node for the whole network: forward = kernel sequence, backward = hand-written kernel sequence"""

    @staticmethod
    def forward(fctx, model, hidden_states, enc, pooled, timestep, guidance, img_ids, txt_ids, key_bias, *lora_params):
        out, ctx = model._engine_forward(hidden_states.detach().to(BF16), enc.detach().to(BF16), pooled.detach(), timestep.detach(),
                                         None if guidance is None else guidance.detach(), img_ids, txt_ids, save=True, key_bias=key_bias)
        fctx.model, fctx.ectx, fctx.n_lora = model, ctx, len(lora_params)
        return out

    @staticmethod
    def backward(fctx, dout):
        model = fctx.model
        if model.grad_sync is not None:
            model.grad_sync.begin()
        model._engine_backward(fctx.ectx, dout)
        fctx.ectx = None
        if model.grad_sync is not None:
            model.grad_scale_from_sync = model.grad_sync.finish()   # all slices reduced (SUM); optimizer folds 1/world
        # hand autograd a private flat copy (one 4 B/param copy) so .grad never aliases the arena the next backward overwrites;
        # the per-parameter grads stay views of ONE contiguous buffer, which the fused optimizer / RCCL all-reduce exploit.
        from ..training.grad_sync import hand_over_gradients
        gflat = hand_over_gradients(model, model.lora_grad_flat)
        model._last_grad_flat = gflat
        grads, off = [], 0
        for p in model._lora_params:
            n = p.numel()
            grads.append(gflat[off:off + n].view_as(p))
            off += n
        return (None,) * 9 + tuple(grads)


class _FluxFullFn(torch.autograd.Function):
    """full-rank training: one autograd node; backward fills the bf16 gradient arena and hands autograd views of a private copy"""

    @staticmethod
    def forward(fctx, model, hidden_states, enc, pooled, timestep, guidance, img_ids, txt_ids, key_bias, *params):
        out, ctx = model._engine_forward(hidden_states.detach().to(BF16), enc.detach().to(BF16), pooled.detach(), timestep.detach(),
                                         None if guidance is None else guidance.detach(), img_ids, txt_ids, save=True, key_bias=key_bias, full=True)
        fctx.model, fctx.ectx = model, ctx
        return out

    @staticmethod
    def backward(fctx, dout):
        model = fctx.model
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
        return (None,) * 9 + tuple(grads)
