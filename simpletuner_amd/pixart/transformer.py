"""PixArt-Sigma DiT and its ControlNet-Transformer wrapper on the MI355X path.

Mirrors the reference's module surface: `PixArtTransformer2DModel` (simpletuner/helpers/models/pixart/transformer.py:146-853: same constructor
arguments and diffusers state-dict keys, `forward(hidden_states, encoder_hidden_states, timestep, added_cond_kwargs, encoder_attention_mask,
return_dict)`), `PixArtSigmaControlNetAdapterModel` / `PixArtSigmaControlNetTransformerModel` (pixart/controlnet.py:17-326: zero-init `before_proj`
on the first copied block, copied blocks 0..N-1, zero-init `after_proj`, residual into the trunk BEFORE trunk blocks 1..N, `from_transformer`).
BASELINE.json configs[4] trains the ControlNet branch only: the trunk is frozen (forward + input gradients), the adapter is a full fine-tune
(weight gradients through the TN GEMM into one bf16 gradient arena -> one fused optimizer launch).

Kernels: the MMDiT set (AdaLN modulate, GEMM with GELU / gate-residual epilogues, flash attention fwd/bwd) + the cross-attention entry points with
the additive key bias of the text mask.  head_dim 72 is not an MFMA-friendly width: the q/k/v projections are stored with each head zero-padded to
96 channels = three 32-row MFMA tiles (and the out-projections with the matching zero K columns), and the attention kernels are instantiated for
head_dim 96 (192-byte LDS rows with a 2-bit XOR swizzle) with scale 1/sqrt(72): 1.33x the attention FLOPs of an exact-72 kernel.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import ops
from ..flux.transformer import LoraGroup, _attach, _frozen
from ..ops import EPI_ADD, EPI_GATE_RESIDUAL, EPI_GELU, EPI_MUL_GELU_GRAD, EPI_NONE
from ..training.checkpoint_plan import CheckpointPlanMixin

BF16 = torch.bfloat16
F32 = torch.float32
_BLOCK_ABI = __import__("os").environ.get("ST355_BLOCK_ABI", "1") != "0"      # A/B switch: 0 = sequence the blocks' kernels from the host instead of st355_block_pixart_*
HP = 96           # padded head width (72 -> 96: three 32-row MFMA tiles; was 128 before the head_dim-96 kernels existed)


def sincos_2d_hw(embed_dim: int, h: int, w: int, base_size: int, interpolation_scale: float) -> torch.Tensor:
    """diffusers get_2d_sincos_pos_embed on a (h, w) grid (first half from the w coordinate; sin then cos)"""
    gh = (torch.arange(h, dtype=torch.float32) / (h / base_size) / interpolation_scale).double()
    gw = (torch.arange(w, dtype=torch.float32) / (w / base_size) / interpolation_scale).double()
    cw = gw[None, :].expand(h, w).reshape(-1)
    chh = gh[:, None].expand(h, w).reshape(-1)

    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
        out = pos[:, None] * omega[None, :]
        return torch.cat([out.sin(), out.cos()], dim=1)

    return torch.cat([one_d(embed_dim // 2, cw), one_d(embed_dim // 2, chh)], dim=1).float()


def _p64(t):
    r = t.shape[0]
    if r % 64 == 0 and t.is_contiguous():
        return t
    o = torch.zeros((r + 63) // 64 * 64, t.shape[1], dtype=BF16, device=t.device)
    o[:r] = t
    return o


class _Block:
    """one BasicTransformerBlock(ada_norm_single): true parameters (diffusers shapes, views of an arena) + the head-padded working copies"""

    def __init__(self, owner, prefix: str, D: int, H: int, hd: int, Dc: int, alloc, trainable: bool):
        self.prefix, self.D, self.H, self.hd, self.trainable = prefix, D, H, hd, trainable
        P = {}
        for nm, shape in (("attn1.to_q", (D, D)), ("attn1.to_k", (D, D)), ("attn1.to_v", (D, D)), ("attn1.to_out.0", (D, D)),
                          ("attn2.to_q", (D, D)), ("attn2.to_k", (D, Dc)), ("attn2.to_v", (D, Dc)), ("attn2.to_out.0", (D, D)),
                          ("ff.net.0.proj", (4 * D, D)), ("ff.net.2", (D, 4 * D))):
            P[nm + ".weight"] = alloc(prefix + nm + ".weight", shape)
            P[nm + ".bias"] = alloc(prefix + nm + ".bias", (shape[0],))
        P["scale_shift_table"] = alloc(prefix + "scale_shift_table", (6, D))
        self.P = P
        self.G: Dict[str, torch.Tensor] = {}          # gradient views (trainable blocks)
        self.W = None                                  # padded working copies
        self.fp8 = False                               # frozen blocks only: fp8-native Linears (fp8_native.py:25-119)
        self.lora = None                               # LoRA adapter groups of this block's attention projections (PixArtTransformer2DModel.add_lora_adapter)

    @torch.no_grad()
    def refresh(self, dev):
        """head-padded working weights: q/k/v rows of head h at [h*128, h*128+72), out-projection columns likewise; + K-major copies for dgrad"""
        P, D, H, hd = self.P, self.D, self.H, self.hd
        Dp = H * HP

        def pad_rows(ws, bs):                                        # list of [D, K] -> [len*Dp, K]
            K = ws[0].shape[1]
            w = torch.zeros(len(ws), H, HP, K, dtype=BF16, device=dev)
            b = torch.zeros(len(ws), H, HP, dtype=BF16, device=dev)
            for j, (wj, bj) in enumerate(zip(ws, bs)):
                w[j, :, :hd] = wj.view(H, hd, K)
                b[j, :, :hd] = bj.view(H, hd)
            return w.view(len(ws) * Dp, K), b.view(len(ws) * Dp)

        def pad_cols(w):                                             # [N, D] -> [N, Dp]
            o = torch.zeros(w.shape[0], H, HP, dtype=BF16, device=dev)
            o[:, :, :hd] = w.view(w.shape[0], H, hd)
            return o.view(w.shape[0], Dp)

        W = SimpleNamespace()
        W.qkv_w, W.qkv_b = pad_rows([P[f"attn1.to_{n}.weight"] for n in "qkv"], [P[f"attn1.to_{n}.bias"] for n in "qkv"])
        W.out1_w, W.out1_b = pad_cols(P["attn1.to_out.0.weight"]), P["attn1.to_out.0.bias"]
        W.q2_w, W.q2_b = pad_rows([P["attn2.to_q.weight"]], [P["attn2.to_q.bias"]])
        W.kv2_w, W.kv2_b = pad_rows([P["attn2.to_k.weight"], P["attn2.to_v.weight"]], [P["attn2.to_k.bias"], P["attn2.to_v.bias"]])
        W.out2_w, W.out2_b = pad_cols(P["attn2.to_out.0.weight"]), P["attn2.to_out.0.bias"]
        W.ff1_w, W.ff1_b, W.ff2_w, W.ff2_b = P["ff.net.0.proj.weight"], P["ff.net.0.proj.bias"], P["ff.net.2.weight"], P["ff.net.2.bias"]
        for n in ("qkv", "out1", "q2", "kv2", "out2", "ff1", "ff2"):
            w = getattr(W, n + "_w")
            if self.fp8:
                # fp8-native base weights (quantize_weight_to_fp8: e4m3 with one scale per output row); the input gradient uses the DEQUANTISED
                # weight exactly like _Fp8NativeLinearFn.backward (fp8_native.py:104-111)
                q, sc = ops.fp8_quantize_weight(w.contiguous())
                setattr(W, n + "_q", q); setattr(W, n + "_s", sc)
                w = (q.view(torch.float8_e4m3fn).to(BF16) * sc.to(BF16).unsqueeze(1))
                setattr(W, n + "_w", None)
            if n != "kv2":
                setattr(W, n + "_wT", w.t().contiguous())
        self.W = W


class PixArtTransformer2DModel(nn.Module):
    def __init__(self, num_attention_heads: int = 16, attention_head_dim: int = 72, in_channels: int = 4, out_channels: Optional[int] = 8,
                 num_layers: int = 28, cross_attention_dim: Optional[int] = 1152, sample_size: int = 128, patch_size: int = 2,
                 interpolation_scale: Optional[float] = None, use_additional_conditions: Optional[bool] = None, caption_channels: Optional[int] = 4096,
                 norm_type: str = "ada_norm_single", device=None, fp8_base: bool = False, **_ignored):
        super().__init__()
        self.fp8_base = bool(fp8_base)             # base_model_precision = fp8 (fp8_native): every Linear of the frozen blocks runs on the fp8 MFMA path
        if norm_type != "ada_norm_single" or patch_size != 2:
            raise NotImplementedError("PixArt(st355): ada_norm_single blocks with patch_size 2 only")
        H, hd = num_attention_heads, attention_head_dim
        D = H * hd
        # head_dim-96 attention contract (include/st355.h): a 96-wide head is a zero-padded narrower one and the 64-row bodies contract q.k / dO.v over
        # 80 channels only — a real head wider than 80 would silently lose channels [80, 96) on those paths, so it is refused here (PixArt heads are 72)
        if hd > 80 or D % 64 or (cross_attention_dim or D) != D or caption_channels is None or caption_channels % 64:
            raise ValueError("PixArt(st355): head_dim <= 80 (zero-padded to 96), inner dim a multiple of 64, cross_attention_dim == inner dim, caption_channels % 64 == 0")
        if use_additional_conditions is None:
            use_additional_conditions = sample_size == 128
        out_channels = in_channels if out_channels is None else out_channels
        self.config = SimpleNamespace(num_attention_heads=H, attention_head_dim=hd, in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                                      cross_attention_dim=D, sample_size=sample_size, patch_size=patch_size, caption_channels=caption_channels,
                                      interpolation_scale=interpolation_scale if interpolation_scale is not None else max(sample_size // 64, 1),
                                      use_additional_conditions=use_additional_conditions, norm_type=norm_type)
        self.inner_dim, self.H, self.hd, self.out_channels = D, H, hd, out_channels
        self.use_additional_conditions = use_additional_conditions
        self.device_ = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        dev = self.device_
        self._specs = []

        def alloc(name, shape):
            t = torch.zeros(*shape, dtype=BF16, device=dev)
            _attach(self, name, _frozen(t))
            return t

        self.P = {}
        for nm, shape in (("pos_embed.proj.weight", (D, in_channels, 2, 2)), ("pos_embed.proj.bias", (D,)),
                          ("adaln_single.emb.timestep_embedder.linear_1.weight", (D, 256)), ("adaln_single.emb.timestep_embedder.linear_1.bias", (D,)),
                          ("adaln_single.emb.timestep_embedder.linear_2.weight", (D, D)), ("adaln_single.emb.timestep_embedder.linear_2.bias", (D,)),
                          ("adaln_single.linear.weight", (6 * D, D)), ("adaln_single.linear.bias", (6 * D,)),
                          ("caption_projection.linear_1.weight", (D, caption_channels)), ("caption_projection.linear_1.bias", (D,)),
                          ("caption_projection.linear_2.weight", (D, D)), ("caption_projection.linear_2.bias", (D,)),
                          ("scale_shift_table", (2, D)), ("proj_out.weight", (4 * out_channels, D)), ("proj_out.bias", (4 * out_channels,))):
            self.P[nm] = alloc(nm, shape)
        if use_additional_conditions:
            sd_ = D // 3
            for e in ("resolution_embedder", "aspect_ratio_embedder"):
                for nm, shape in ((f"adaln_single.emb.{e}.linear_1.weight", (sd_, 256)), (f"adaln_single.emb.{e}.linear_1.bias", (sd_,)),
                                  (f"adaln_single.emb.{e}.linear_2.weight", (sd_, sd_)), (f"adaln_single.emb.{e}.linear_2.bias", (sd_,))):
                    self.P[nm] = alloc(nm, shape)
        self.blocks: List[_Block] = [_Block(self, f"transformer_blocks.{i}.", D, H, hd, D, alloc, trainable=False) for i in range(num_layers)]
        self._prepared = False
        self._cache: Dict = {}
        self.lora_groups: List[LoraGroup] = []
        self._lora_params: List[nn.Parameter] = []
        self.accumulate_lora_grads = False
        self.grad_sync = None
        self._tread_router, self._tread_routes = None, None

    # ---- LoRA (peft naming; pixart/model.py:59 DEFAULT_LORA_TARGET = to_k, to_q, to_v, to_out.0: attn1 and attn2 of every block) ----
    def add_lora_adapter(self, rank: int = 32, alpha: Optional[float] = None, seed: int = 7, init_b_std: float = 0.0):
        """Adapters on attn1 / attn2 to_q, to_k, to_v, to_out.0 of every trunk block, riding in the K-extension of the projections' GEMMs.  The working layout pads
        every head from 72 to 96 lanes, so the adapter factors live in that layout too: lora_B of to_q / to_k / to_v has H * 96 rows and lora_A of to_out.0 has
        H * 96 columns, the pad rows / columns initialised to ZERO.  They stay zero: the gradient of a pad row of B is dY_pad^T T and dY is exactly zero on the pad
        lanes (K / Q / W_out pad lanes are zero), the gradient of a pad column of A is U^T O_pad with O_pad = 0 — and AdamW leaves a zero parameter with a zero
        gradient at zero.  `lora_state_dict()` hands out the true-shaped (peft) tensors."""
        # over an fp8-native trunk (base_model_precision fp8 + adapters: the reference's published sweeps carry such rows, SEGMENTED_CHECKPOINTING.md:786,819,846) the
        # base Linear runs on the fp8 pipe and the low-rank term is added in bf16 (peft's LoraLayer around Fp8NativeLinear): `_block_fwd_fp8`
        alpha = float(rank if alpha is None else alpha)
        D, H, hd, dev = self.inner_dim, self.H, self.hd, self.device_
        Dp = H * HP
        plan = []
        self.lora_groups = []
        for i, blk in enumerate(self.blocks):
            p = f"transformer_blocks.{i}."
            mk = lambda K, N, targets: LoraGroup(K, N, targets, rank, alpha, dev)
            blk.lora = SimpleNamespace(
                qkv=mk(D, 3 * Dp, [(p + "attn1.to_q", 0, Dp), (p + "attn1.to_k", Dp, Dp), (p + "attn1.to_v", 2 * Dp, Dp)]),
                out1=mk(Dp, D, [(p + "attn1.to_out.0", 0, D)]),
                q2=mk(D, Dp, [(p + "attn2.to_q", 0, Dp)]),
                kv2=mk(D, 2 * Dp, [(p + "attn2.to_k", 0, Dp), (p + "attn2.to_v", Dp, Dp)]),
                out2=mk(Dp, D, [(p + "attn2.to_out.0", 0, D)]))
            for g in (blk.lora.qkv, blk.lora.out1, blk.lora.q2, blk.lora.kv2, blk.lora.out2):
                self.lora_groups.append(g)
                for (name, _, N) in g.targets:
                    plan.append((g, name, N, g.K))
        total = (sum(rank * K + N * rank for (_, _, N, K) in plan) + 7) // 8 * 8
        self.lora_flat = torch.zeros(total, dtype=F32, device=dev)
        self.lora_grad_flat = torch.zeros(total, dtype=F32, device=dev)
        gen = torch.Generator(device=dev).manual_seed(seed)
        off = 0
        self._lora_params = []
        pad_rows = torch.zeros(H, HP, dtype=torch.bool, device=dev); pad_rows[:, hd:] = True          # the pad lanes of a head-padded axis
        pad_rows = pad_rows.reshape(-1)
        for (g, name, N, K) in plan:
            if not g.A:
                g.flat_lo = off
            a = self.lora_flat[off:off + rank * K].view(rank, K); ga = self.lora_grad_flat[off:off + rank * K].view(rank, K)
            off += rank * K
            b = self.lora_flat[off:off + N * rank].view(N, rank); gb = self.lora_grad_flat[off:off + N * rank].view(N, rank)
            off += N * rank
            g.flat_hi = off
            k_true = D                                                           # kaiming_uniform(a=sqrt(5)) on the TRUE [r, in_features] (peft default for lora_A)
            a.copy_((torch.rand(rank, K, generator=gen, device=dev) * 2 - 1) * (1.0 / math.sqrt(k_true)))
            if K == Dp:
                a[:, pad_rows] = 0
            if init_b_std > 0:
                b.copy_(torch.randn(N, rank, generator=gen, device=dev) * init_b_std)
                if N == Dp:
                    b[pad_rows] = 0
            pa, pb = nn.Parameter(a), nn.Parameter(b)
            _attach(self, name + ".lora_A.default.weight", pa); _attach(self, name + ".lora_B.default.weight", pb)
            g.A.append(pa.data); g.B.append(pb.data); g.gA.append(ga); g.gB.append(gb)
            self._lora_params += [pa, pb]
        return self._lora_params

    def lora_state_dict(self) -> Dict[str, torch.Tensor]:
        """the adapters in their TRUE (peft) shapes: the pad lanes of the head-padded axes dropped"""
        H, hd, Dp = self.H, self.hd, self.H * HP
        out = {}
        for name, p in self.named_parameters():
            if ".lora_A." in name and p.shape[1] == Dp:
                out[name] = p.detach().view(p.shape[0], H, HP)[:, :, :hd].reshape(p.shape[0], H * hd).clone()
            elif ".lora_B." in name and p.shape[0] == Dp:
                out[name] = p.detach().view(H, HP, p.shape[1])[:, :hd].reshape(H * hd, p.shape[1]).clone()
            elif ".lora_" in name:
                out[name] = p.detach().clone()
        return out

    @torch.no_grad()
    def load_lora_state_dict(self, state: Dict[str, torch.Tensor]):
        """true-shaped (peft) adapter tensors, keyed `<module>.lora_A[.default].weight`, into the head-padded working layout (pad lanes zero)"""
        H, hd, Dp = self.H, self.hd, self.H * HP
        norm = {k.replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B."): v for k, v in state.items()}
        for name, p in self.named_parameters():
            if ".lora_" not in name:
                continue
            v = norm[name.replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B.")].to(device=p.device, dtype=p.dtype)
            if ".lora_A." in name and p.shape[1] == Dp:
                p.zero_(); p.view(p.shape[0], H, HP)[:, :, :hd].copy_(v.view(p.shape[0], H, hd))
            elif ".lora_B." in name and p.shape[0] == Dp:
                p.zero_(); p.view(H, HP, p.shape[1])[:, :hd].copy_(v.view(H, hd, p.shape[1]))
            else:
                p.copy_(v)

    def set_router(self, router, routes):
        """pixart/transformer.py:487-489: TREAD router + [{selection_ratio, start_layer_idx, end_layer_idx}] (training/tread.py)"""
        self._tread_router, self._tread_routes = router, routes

    # ---- weights ----
    @torch.no_grad()
    def init_synthetic(self, seed: int = 42):
        g = torch.Generator(device=self.device_).manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("scale_shift_table"):
                p.data.copy_(torch.randn(p.shape, generator=g, device=self.device_) / math.sqrt(self.inner_dim))
            elif name.endswith(".bias"):
                p.data.copy_(0.02 * torch.randn(p.shape, generator=g, device=self.device_))
            else:
                p.data.copy_(torch.randn(p.shape, generator=g, device=self.device_) / math.sqrt(p[0].numel()))
        self._prepared = False

    @torch.no_grad()
    def load_flat_state(self, state: Dict[str, torch.Tensor]):
        own = dict(self.named_parameters())
        missing = [k for k in own if k not in state]
        if missing:
            raise KeyError(f"missing weights: {missing[:5]} ... ({len(missing)})")
        for k, v in state.items():
            if k in own:
                own[k].data.copy_(v.to(device=own[k].device, dtype=own[k].dtype))
        self._prepared = False

    @torch.no_grad()
    def prepare(self):
        for b in self.blocks:
            b.fp8 = self.fp8_base
            b.refresh(self.device_)
        P = self.P
        self.patch_w = P["pos_embed.proj.weight"].reshape(self.inner_dim, -1)
        K = self.patch_w.shape[1]
        if K % 64:                                                   # 4 latent channels * 4 = 16 -> K granule 64
            w = torch.zeros(self.inner_dim, 64, dtype=BF16, device=self.device_); w[:, :K] = self.patch_w; self.patch_w = w
        n_out = P["proj_out.weight"].shape[0]
        self.proj_out_wT = torch.zeros(self.inner_dim, (n_out + 63) // 64 * 64, dtype=BF16, device=self.device_)
        self.proj_out_wT[:, :n_out] = P["proj_out.weight"].t()
        self._prepared = True

    # ---- shared pieces ----
    def _pos(self, h, w, B):
        key = (h, w, B)
        hit = self._cache.get(key)
        if hit is None:
            c = self.config
            pos = sincos_2d_hw(self.inner_dim, h, w, c.sample_size // c.patch_size, c.interpolation_scale).to(self.device_, BF16)
            hit = self._cache[key] = pos[None].expand(B, -1, -1).reshape(B * h * w, -1).contiguous()
        return hit

    def _patch_embed(self, x):
        B, C, Hh, Ww = x.shape
        pt = ops.patchify(x.to(BF16).contiguous(), order=0).view(B * (Hh // 2) * (Ww // 2), 4 * C)
        if pt.shape[1] != self.patch_w.shape[1]:
            pp = torch.zeros(pt.shape[0], self.patch_w.shape[1], dtype=BF16, device=pt.device); pp[:, :pt.shape[1]] = pt; pt = pp
        return ops.gemm(pt, self.patch_w, bias=self.P["pos_embed.proj.bias"], epilogue=EPI_ADD, aux_in=self._pos(Hh // 2, Ww // 2, B))

    def _mlp(self, x, prefix):
        P = self.P
        return ops.gemm(ops.silu(ops.gemm(x, P[prefix + ".linear_1.weight"], bias=P[prefix + ".linear_1.bias"])), P[prefix + ".linear_2.weight"],
                        bias=P[prefix + ".linear_2.bias"])

    def _conditioning(self, timestep, added_cond_kwargs, B, Hh, Ww):
        """AdaLayerNormSingle: (t6 [B,6D], emb [B,D]).  TOKENWISE timesteps [B, S] (CREPA self-flow; pixart/transformer.py:790-850 `_embed_timesteps`): one row per token,
        (t6 [B*S,6D], emb [B*S,D]) — the blocks and the head then index their modulation rows per token (rows_per_batch = 1)"""
        dev = self.device_
        tok = timestep.dim() == 2
        S_tok = (Hh // 2) * (Ww // 2)
        if tok and tuple(timestep.shape) != (B, S_tok):
            raise ValueError(f"PixArt tokenwise timestep embedding expected shape ({B}, {S_tok}), got {tuple(timestep.shape)}.")       # pixart/transformer.py:807-810
        t32 = timestep.to(device=dev, dtype=F32).reshape(-1).contiguous() if tok else timestep.to(device=dev, dtype=F32).reshape(-1).expand(B).contiguous()
        emb = self._mlp(ops.timestep_proj(t32, 256, 1.0), "adaln_single.emb.timestep_embedder")
        if self.use_additional_conditions:
            ack = added_cond_kwargs or {}
            res = ack.get("resolution")
            ar = ack.get("aspect_ratio")
            if res is None:
                res = torch.tensor([[Hh, Ww]], device=dev, dtype=F32).expand(B, -1)
            if ar is None:
                ar = torch.tensor([[float(Hh / Ww)]], device=dev, dtype=F32).expand(B, -1)
            r = self._mlp(ops.timestep_proj(res.to(device=dev, dtype=F32).reshape(-1).contiguous(), 256, 1.0), "adaln_single.emb.resolution_embedder").reshape(B, -1)
            a = self._mlp(ops.timestep_proj(ar.to(device=dev, dtype=F32).reshape(-1).contiguous(), 256, 1.0), "adaln_single.emb.aspect_ratio_embedder").reshape(B, -1)
            size = torch.cat([r, a], dim=1).contiguous()
            if tok:         # the size conditions are shared by a sample's tokens (:840-843)
                size = size[:, None, :].expand(B, S_tok, size.shape[1]).reshape(B * S_tok, -1)
            emb = ops.add(emb, size)
        t6 = ops.gemm(ops.silu(emb), self.P["adaln_single.linear.weight"], bias=self.P["adaln_single.linear.bias"])
        return t6, emb

    def _caption(self, enc):
        B, Sk, _ = enc.shape
        P = self.P
        e = ops.gemm(enc.to(BF16).reshape(B * Sk, -1).contiguous(), P["caption_projection.linear_1.weight"], bias=P["caption_projection.linear_1.bias"],
                     epilogue=EPI_GELU)
        return ops.gemm(e, P["caption_projection.linear_2.weight"], bias=P["caption_projection.linear_2.bias"])

    # ---- one block: forward (optionally saving) and backward ----
    def _block_fwd(self, blk: _Block, h, ctx2d, kbias, t6, B, S, Sk, save: bool, exact: bool = False):
        """`exact`: a forward whose activations are NOT kept but whose values must equal the keeping forward's bit for bit (the first pass over a checkpointed
        segment): the GELU epilogue then also rounds its pre-activation through a (discarded) bf16 buffer, as the keeping form does"""
        D, H, W = self.inner_dim, self.H, blk.W
        Dp = H * HP
        scale = 1.0 / math.sqrt(self.hd)
        mod = (blk.P["scale_shift_table"].view(1, 6 * D) + t6).contiguous()             # [B, 6D] (tiny); tokenwise timesteps: [B*S, 6D], one row per token
        m = [mod[:, k * D:(k + 1) * D] for k in range(6)]                               # shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        rpb = (B * S) // mod.shape[0]                                                   # rows that share a modulation row: S, or 1 (tokenwise)
        train = save and blk.trainable
        if rpb != S and (train or blk.fp8):
            raise NotImplementedError("PixArt tokenwise timesteps: the frozen bf16 trunk forward only (the reference's ControlNet wrapper takes per-sample timesteps, "
                                      "pixart/controlnet.py:241-246; the fp8 trunk form is not built for them)")
        if blk.fp8:
            return self._block_fwd_fp8(blk, h, ctx2d, kbias, mod, m, B, S, Sk, save)
        L = blk.lora
        if L is not None and rpb != S:
            raise NotImplementedError("PixArt LoRA: per-sample timesteps only")
        kx = (lambda g, T: dict(a2=T, b2=g.B_blk, k2_real=g.k2_real)) if L is not None else (lambda g, T: {})
        if _BLOCK_ABI and ops.ATTN_TR and h.is_contiguous() and ctx2d.is_contiguous() and rpb == S and L is None:
            # the block as ONE C entry point (st355_block_pixart_fwd): the launches of the host-side sequencing below, in its order, on its
            # operands (every buffer allocated here, kept ones handed to the backward as before) — bit-identical to it (ST355_BLOCK_ABI=0 restores it)
            dev = h.device
            Sp, Skp = (S + 63) // 64 * 64, (Sk + 63) // 64 * 64
            e = lambda *sh, dt=BF16: torch.empty(*sh, dtype=dt, device=dev)
            n1, qkv, O, h1, q2, kv, O2, h2, n2, a, h3 = (e(B * S, D), e(B * S, 3 * Dp), e(B * S, Dp), e(B * S, D), e(B * S, Dp), e(B * Sk, 2 * Dp), e(B * S, Dp),
                                                         e(B * S, D), e(B * S, D), e(B * S, 4 * D), e(B * S, D))
            Q, K, Q2, K2 = e(B, H, S, HP), e(B, H, S, HP), e(B, H, S, HP), e(B, H, Sk, HP)
            lse, lse2 = e(B, H, S, dt=F32), e(B, H, S, dt=F32)
            Vt = (torch.zeros if Sp > S else torch.empty)(B, H, HP, Sp, dtype=BF16, device=dev)
            V2t = (torch.zeros if Skp > Sk else torch.empty)(B, H, HP, Skp, dtype=BF16, device=dev)
            ya, yf = (e(B * S, D), e(B * S, D)) if train else (None, None)
            pre = e(B * S, 4 * D) if (save or exact) else None
            ops.block_pixart_fwd(B=B, S=S, Sk=Sk, H=H, D=D, d_pad=HP, scale=scale, h=h, ctx=ctx2d, mod=mod, mod_stride=6 * D, key_bias=kbias,
                                 w_qkv=W.qkv_w, b_qkv=W.qkv_b, w_out1=W.out1_w, b_out1=W.out1_b, w_q2=W.q2_w, b_q2=W.q2_b, w_kv2=W.kv2_w, b_kv2=W.kv2_b,
                                 w_out2=W.out2_w, b_out2=W.out2_b, w_ff1=W.ff1_w, b_ff1=W.ff1_b, w_ff2=W.ff2_w, b_ff2=W.ff2_b,
                                 n1=n1, qkv=qkv, Q=Q, K=K, O=O, lse=lse, ya=ya, h1=h1, q2=q2, kv=kv, Q2=Q2, K2=K2, O2=O2, lse_x=lse2, h2=h2, n2=n2,
                                 pre=pre, act=a, yf=yf, Vt=Vt, V2t=V2t, out=h3)
            sv = None
            if save:
                sv = SimpleNamespace(h=h, mod=mod, m=m, n1=n1, qkv=qkv, Q=Q, Qt=None, K=K, Kt=None, O=O, lse=lse, Sp=Sp, ya=ya, h1=h1, q2=q2, kv=kv, Q2=Q2,
                                     Q2t=None, K2=K2, K2t=None, Skp=Skp, O2=O2, lse2=lse2, h2=h2, n2=n2, pre=pre, a=a, yf=yf)
            return h3, sv
        n1 = ops.ln_modulate_fwd(h, m[1], m[0], rpb)
        T_qkv = ops.gemm(n1, L.qkv.A_cat) if L is not None else None
        qkv = ops.gemm(n1, W.qkv_w, bias=W.qkv_b, **kx(L.qkv if L is not None else None, T_qkv))
        Q, Qt, Sp = ops.head_split(qkv[:, :Dp], B, H, HP, S, want_xt=not ops.ATTN_TR)
        K, Kt, _ = ops.head_split(qkv[:, Dp:2 * Dp], B, H, HP, S, want_xt=not ops.ATTN_TR)
        _, Vt, _ = ops.head_split(qkv[:, 2 * Dp:], B, H, HP, S, want_x=False)
        O = torch.empty(B * S, Dp, dtype=BF16, device=h.device)
        lse = torch.empty(B, H, S, dtype=F32, device=h.device)
        ops.attn_fwd(Q, K, Vt, O, lse, B, H, S, Sp, HP, scale)
        ya = torch.empty(B * S, D, dtype=BF16, device=h.device) if train else None
        T_o1 = ops.gemm(O, L.out1.A_cat) if L is not None else None
        h1 = ops.gemm(O, W.out1_w, bias=W.out1_b, epilogue=EPI_GATE_RESIDUAL, gate=m[2], aux_in=h, rows_per_batch=rpb, aux_out=ya, **kx(L.out1 if L is not None else None, T_o1))
        T_q2 = ops.gemm(h1, L.q2.A_cat) if L is not None else None
        q2 = ops.gemm(h1, W.q2_w, bias=W.q2_b, **kx(L.q2 if L is not None else None, T_q2))
        T_kv = ops.gemm(ctx2d, L.kv2.A_cat) if L is not None else None
        kv = ops.gemm(ctx2d, W.kv2_w, bias=W.kv2_b, **kx(L.kv2 if L is not None else None, T_kv))
        Q2, Q2t, _ = ops.head_split(q2, B, H, HP, S, want_xt=not ops.ATTN_TR)
        K2, K2t, Skp = ops.head_split(kv[:, :Dp], B, H, HP, Sk, want_xt=not ops.ATTN_TR)
        _, V2t, _ = ops.head_split(kv[:, Dp:], B, H, HP, Sk, want_x=False)
        O2 = torch.empty(B * S, Dp, dtype=BF16, device=h.device)
        lse2 = torch.empty(B, H, S, dtype=F32, device=h.device)
        ops.attn_cross_fwd(Q2, K2, V2t, O2, lse2, B, H, S, Sk, Skp, HP, scale, key_bias=kbias)
        T_o2 = ops.gemm(O2, L.out2.A_cat) if L is not None else None
        h2 = ops.gemm(O2, W.out2_w, bias=W.out2_b, epilogue=EPI_ADD, aux_in=h1, **kx(L.out2 if L is not None else None, T_o2))
        n2 = ops.ln_modulate_fwd(h2, m[4], m[3], rpb)
        pre = torch.empty(B * S, 4 * D, dtype=BF16, device=h.device) if (save or exact) else None
        a = ops.gemm(n2, W.ff1_w, bias=W.ff1_b, epilogue=EPI_GELU, aux_out=pre)
        yf = torch.empty(B * S, D, dtype=BF16, device=h.device) if train else None
        h3 = ops.gemm(a, W.ff2_w, bias=W.ff2_b, epilogue=EPI_GATE_RESIDUAL, gate=m[5], aux_in=h2, rows_per_batch=rpb, aux_out=yf)
        sv = None
        if save:
            sv = SimpleNamespace(h=h, mod=mod, m=m, n1=n1, qkv=qkv, Q=Q, Qt=Qt, K=K, Kt=Kt, O=O, lse=lse, Sp=Sp, ya=ya, h1=h1, q2=q2, kv=kv, Q2=Q2, Q2t=Q2t, K2=K2,
                                 K2t=K2t, Skp=Skp, O2=O2, lse2=lse2, h2=h2, n2=n2, pre=pre, a=a, yf=yf, T_qkv=T_qkv, T_o1=T_o1, T_q2=T_q2, T_kv=T_kv, T_o2=T_o2)
        return h3, sv

    def _block_fwd_fp8(self, blk: _Block, h, ctx2d, kbias, mod, m, B, S, Sk, save: bool):
        """the same block with every Linear in the reference's fp8-native form: e5m2 activations (one scale per call), e4m3 weights (row scales),
        fp8 MFMA with fp32 accumulation, bf16 out; activation functions / gates / residuals are then passes of their own, as in the reference
        (where they are separate torch ops around Fp8NativeLinear).  Frozen blocks only (the backward needs input gradients, computed with the
        dequantised weights)."""
        D, H, W = self.inner_dim, self.H, blk.W
        Dp = H * HP
        scale = 1.0 / math.sqrt(self.hd)

        L = blk.lora
        Ts = {}

        def lin8(x, name, bias):
            xq, sa = ops.fp8_quantize_act(x)
            y = ops.linear_fp8(xq, sa, getattr(W, name + "_q"), getattr(W, name + "_s"), bias=bias)
            g = getattr(L, name, None) if L is not None else None
            if g is not None:             # peft's LoraLayer around the fp8-native base Linear: y + s B (A x), the low-rank term in bf16 on the un-quantised input
                T = Ts[name] = ops.gemm(x, g.A_cat)
                y = ops.gemm(T, g.B_blk, epilogue=EPI_ADD, aux_in=y)
            return y

        n1 = ops.ln_modulate_fwd(h, m[1], m[0], S)
        qkv = lin8(n1, "qkv", W.qkv_b)
        Q, Qt, Sp = ops.head_split(qkv[:, :Dp], B, H, HP, S, want_xt=not ops.ATTN_TR)
        K, Kt, _ = ops.head_split(qkv[:, Dp:2 * Dp], B, H, HP, S, want_xt=not ops.ATTN_TR)
        _, Vt, _ = ops.head_split(qkv[:, 2 * Dp:], B, H, HP, S, want_x=False)
        O = torch.empty(B * S, Dp, dtype=BF16, device=h.device)
        lse = torch.empty(B, H, S, dtype=F32, device=h.device)
        ops.attn_fwd(Q, K, Vt, O, lse, B, H, S, Sp, HP, scale)
        h1 = ops.add(h, ops.scale_cols(lin8(O, "out1", W.out1_b), m[2], S))
        q2 = lin8(h1, "q2", W.q2_b)
        kv = lin8(ctx2d, "kv2", W.kv2_b)
        Q2, Q2t, _ = ops.head_split(q2, B, H, HP, S, want_xt=not ops.ATTN_TR)
        K2, K2t, Skp = ops.head_split(kv[:, :Dp], B, H, HP, Sk, want_xt=not ops.ATTN_TR)
        _, V2t, _ = ops.head_split(kv[:, Dp:], B, H, HP, Sk, want_x=False)
        O2 = torch.empty(B * S, Dp, dtype=BF16, device=h.device)
        lse2 = torch.empty(B, H, S, dtype=F32, device=h.device)
        ops.attn_cross_fwd(Q2, K2, V2t, O2, lse2, B, H, S, Sk, Skp, HP, scale, key_bias=kbias)
        h2 = ops.add(h1, lin8(O2, "out2", W.out2_b))
        n2 = ops.ln_modulate_fwd(h2, m[4], m[3], S)
        pre = lin8(n2, "ff1", W.ff1_b)
        a = ops.gelu_tanh(pre)
        h3 = ops.add(h2, ops.scale_cols(lin8(a, "ff2", W.ff2_b), m[5], S))
        sv = None
        if save:
            sv = SimpleNamespace(h=h, mod=mod, m=m, n1=n1, qkv=qkv, Q=Q, Qt=Qt, K=K, Kt=Kt, O=O, lse=lse, Sp=Sp, ya=None, h1=h1, q2=q2, kv=kv, Q2=Q2, Q2t=Q2t, K2=K2,
                                 K2t=K2t, Skp=Skp, O2=O2, lse2=lse2, h2=h2, n2=n2, pre=pre, a=a, yf=None, T_qkv=Ts.get("qkv"), T_o1=Ts.get("out1"), T_q2=Ts.get("q2"),
                                 T_kv=Ts.get("kv2"), T_o2=Ts.get("out2"))
        return h3, sv

    def _block_bwd(self, blk: _Block, sv, d3, ctx2d, kbias, B, S, Sk):
        """d3 = dL/d(block output) -> dL/d(block input); fills blk.G for trainable blocks"""
        D, H, W = self.inner_dim, self.H, blk.W
        Dp = H * HP
        scale = 1.0 / math.sqrt(self.hd)
        m = sv.m
        tr = blk.trainable
        dmod = torch.zeros(B, 6 * D, dtype=F32, device=d3.device) if tr else None

        def wgrad(name, dy, x, unpad):
            if not tr:
                return
            gw = ops.gemm_tn(_p64(dy), _p64(x))
            tb = torch.empty(1, dy.shape[1], dtype=F32, device=dy.device)
            ops.colsum_prod(dy, tb)
            unpad(gw, tb[0])

        def unpad_rows(names):                       # padded output rows -> the true [D, K] gradients of the listed projections
            def f(gw, gb):
                K = gw.shape[1]
                g4, b3 = gw.view(len(names), H, HP, K), gb.view(len(names), H, HP)
                for j, nm in enumerate(names):
                    blk.G[nm + ".weight"].view(H, self.hd, K).copy_(g4[j, :, :self.hd])
                    blk.G[nm + ".bias"].view(H, self.hd).copy_(b3[j, :, :self.hd])
            return f

        def unpad_cols(nm):                          # padded input columns -> true [N, D]
            def f(gw, gb):
                blk.G[nm + ".weight"].view(gw.shape[0], H, self.hd).copy_(gw.view(gw.shape[0], H, HP)[:, :, :self.hd])
                blk.G[nm + ".bias"].copy_(gb)
            return f

        def plain(nm):
            def f(gw, gb):
                blk.G[nm + ".weight"].copy_(gw); blk.G[nm + ".bias"].copy_(gb)
            return f

        def mod_grads(dn, x_in, k_shift, k_scale):
            """d shift = sum_t dY, d scale = sum_t dY * LN(x); x_in = the LayerNorm's input (LN(x) recomputed: no division by 1 + scale)"""
            if not tr:
                return
            ops.colsum_prod(dn, dmod[:, k_shift * D:(k_shift + 1) * D], rows_per_batch=S)
            ops.colsum_prod(dn, dmod[:, k_scale * D:(k_scale + 1) * D], b=ops.layer_norm_xhat(x_in), rows_per_batch=S)

        L = blk.lora
        acc, sync = self.accumulate_lora_grads, self.grad_sync

        def lora_dgrad(g, dy, wT, x, T, **kw):
            """d x = dy W (+ (dy sB) A) with the adapter's K-extension, and its rank-space gradients (x: the projection's input; None wT: no input gradient wanted)"""
            if g is None:
                return ops.gemm(dy, wT, **kw)
            U = ops.gemm(dy, g.B_blk_T)
            dx = ops.gemm(dy, wT, a2=U, b2=g.A_cat_T, k2_real=g.k2_real, **kw) if wT is not None else None
            g.grads(x, T, dy, U, acc, sync)
            return dx

        if _BLOCK_ABI and ops.ATTN_TR and sv.Qt is None and sv.pre is not None and d3.is_contiguous() and getattr(W, "ff2_wT", None) is not None and L is None:
            # the data path of the backward as ONE C entry point (st355_block_pixart_bwd); every intermediate gradient stays in the buffers allocated here, and a
            # trainable block takes its weight / bias / modulation gradients from them afterwards — the same launches on the same operands as the host-side
            # sequencing below, the weight-gradient launches after the data path instead of between its steps (independent of it: bit-identical results)
            dev = d3.device
            e = lambda *sh: torch.empty(*sh, dtype=BF16, device=dev)
            dyf, dpre, dn2, d2, dO2, dq2, dkv, d1, dya, dO, dqkv, dn1, d0 = (e(B * S, D), e(B * S, 4 * D), e(B * S, D), e(B * S, D), e(B * S, Dp), e(B * S, Dp),
                                                                            e(B * Sk, 2 * Dp), e(B * S, D), e(B * S, D), e(B * S, Dp), e(B * S, 3 * Dp),
                                                                            e(B * S, D), e(B * S, D))
            dQ, dK = e(B, H, S, HP), e(B, H, max(S, Sk), HP)
            ops.block_pixart_bwd(B=B, S=S, Sk=Sk, H=H, D=D, d_pad=HP, scale=scale, h=sv.h, mod=sv.mod, mod_stride=6 * D, key_bias=kbias,
                                 wT_qkv=W.qkv_wT, wT_out1=W.out1_wT, wT_q2=W.q2_wT, wT_out2=W.out2_wT, wT_ff1=W.ff1_wT, wT_ff2=W.ff2_wT,
                                 qkv=sv.qkv, Q=sv.Q, K=sv.K, O=sv.O, lse=sv.lse, q2=sv.q2, kv=sv.kv, Q2=sv.Q2, K2=sv.K2, O2=sv.O2, lse_x=sv.lse2, h2=sv.h2, pre=sv.pre,
                                 d_out=d3, dyf=dyf, dpre=dpre, dn2=dn2, d2=d2, dO2=dO2, dq2=dq2, dkv=dkv, d1=d1, dya=dya, dO=dO, dqkv=dqkv, dn1=dn1, dQ=dQ, dK=dK,
                                 d_in=d0)
            if tr:
                ops.colsum_prod(d3, dmod[:, 5 * D:6 * D], b=sv.yf, rows_per_batch=S)
                wgrad("ff2", dyf, sv.a, plain("ff.net.2"))
                wgrad("ff1", dpre, sv.n2, plain("ff.net.0.proj"))
                mod_grads(dn2, sv.h2, 3, 4)
                wgrad("out2", d2, sv.O2, unpad_cols("attn2.to_out.0"))
                wgrad("q2", dq2, sv.h1, unpad_rows(["attn2.to_q"]))
                wgrad("kv2", dkv, ctx2d, unpad_rows(["attn2.to_k", "attn2.to_v"]))
                ops.colsum_prod(d1, dmod[:, 2 * D:3 * D], b=sv.ya, rows_per_batch=S)
                wgrad("out1", dya, sv.O, unpad_cols("attn1.to_out.0"))
                wgrad("qkv", dqkv, sv.n1, unpad_rows(["attn1.to_q", "attn1.to_k", "attn1.to_v"]))
                mod_grads(dn1, sv.h, 0, 1)
                blk.G["scale_shift_table"].copy_(dmod.sum(0).view(6, D))
            return d0
        # ---- feed-forward ----
        dyf = ops.scale_cols(d3, m[5], S)
        if tr:
            ops.colsum_prod(d3, dmod[:, 5 * D:6 * D], b=sv.yf, rows_per_batch=S)
        wgrad("ff2", dyf, sv.a, plain("ff.net.2"))
        dpre = ops.gemm(dyf, W.ff2_wT, epilogue=EPI_MUL_GELU_GRAD, aux_in=sv.pre)
        wgrad("ff1", dpre, sv.n2, plain("ff.net.0.proj"))
        dn2 = ops.gemm(dpre, W.ff1_wT)
        mod_grads(dn2, sv.h2, 3, 4)
        d2, _ = ops.ln_modulate_bwd(dn2, sv.h2, m[4], S, dres=d3)
        # ---- cross-attention (no pre-norm, no gate) ----
        wgrad("out2", d2, sv.O2, unpad_cols("attn2.to_out.0"))
        dO2 = lora_dgrad(L.out2 if L is not None else None, d2, W.out2_wT, sv.O2, getattr(sv, "T_o2", None))
        dq2 = torch.empty_like(sv.q2)
        dkv = torch.empty_like(sv.kv)
        dQ, dK = torch.empty_like(sv.Q2), torch.empty_like(sv.K2)
        ops.attn_cross_bwd(sv.Q2, sv.K2, sv.Q2t, sv.K2t, sv.kv[:, Dp:], sv.O2, dO2, sv.lse2, dQ, dK, dkv[:, Dp:], B, H, S, sv.Sp, Sk, sv.Skp, HP, scale, key_bias=kbias)
        ops.head_merge(dQ, dq2, B, H, HP, S)
        ops.head_merge(dK, dkv[:, :Dp], B, H, HP, Sk)
        wgrad("q2", dq2, sv.h1, unpad_rows(["attn2.to_q"]))
        wgrad("kv2", dkv, ctx2d, unpad_rows(["attn2.to_k", "attn2.to_v"]))
        d1 = lora_dgrad(L.q2 if L is not None else None, dq2, W.q2_wT, sv.h1, getattr(sv, "T_q2", None), epilogue=EPI_ADD, aux_in=d2)
        if L is not None:
            lora_dgrad(L.kv2, dkv, None, ctx2d, sv.T_kv)                 # the caption projection is frozen: only the adapters' gradients
        # ---- self-attention ----
        dya = ops.scale_cols(d1, m[2], S)
        if tr:
            ops.colsum_prod(d1, dmod[:, 2 * D:3 * D], b=sv.ya, rows_per_batch=S)
        wgrad("out1", dya, sv.O, unpad_cols("attn1.to_out.0"))
        dO = lora_dgrad(L.out1 if L is not None else None, dya, W.out1_wT, sv.O, getattr(sv, "T_o1", None))
        dqkv = torch.empty_like(sv.qkv)
        dQ, dK = torch.empty_like(sv.Q), torch.empty_like(sv.K)
        ops.attn_bwd(sv.Q, sv.K, sv.Qt, sv.Kt, sv.qkv[:, 2 * Dp:], sv.O, dO, sv.lse, dQ, dK, dqkv[:, 2 * Dp:], B, H, S, sv.Sp, HP, scale)
        ops.head_merge(dQ, dqkv[:, :Dp], B, H, HP, S)
        ops.head_merge(dK, dqkv[:, Dp:2 * Dp], B, H, HP, S)
        wgrad("qkv", dqkv, sv.n1, unpad_rows(["attn1.to_q", "attn1.to_k", "attn1.to_v"]))
        dn1 = lora_dgrad(L.qkv if L is not None else None, dqkv, W.qkv_wT, sv.n1, getattr(sv, "T_qkv", None))
        mod_grads(dn1, sv.h, 0, 1)
        d0, _ = ops.ln_modulate_bwd(dn1, sv.h, m[1], S, dres=d1)
        if tr:
            blk.G["scale_shift_table"].copy_(dmod.sum(0).view(6, D))
        return d0

    def _head(self, h, emb, B, hh, ww):
        D = self.inner_dim
        mod = (self.P["scale_shift_table"].view(1, 2 * D) + torch.cat([emb, emb], dim=1)).contiguous()      # (shift, scale); tokenwise: one row per token
        n = ops.ln_modulate_fwd(h, mod[:, D:], mod[:, :D], (B * hh * ww) // mod.shape[0])
        pk = ops.gemm(n, self.P["proj_out.weight"], bias=self.P["proj_out.bias"])
        out = ops.unpatchify(pk.view(B, hh * ww, -1), self.out_channels, 2 * hh, 2 * ww, order=1)
        return out, mod

    def _head_bwd(self, dout, h_final, mod, B, hh, ww):
        D = self.inner_dim
        dpk = ops.patchify(dout.to(BF16).contiguous(), order=1).view(B * hh * ww, -1)
        dp = torch.zeros(dpk.shape[0], self.proj_out_wT.shape[1], dtype=BF16, device=dpk.device)
        dp[:, :dpk.shape[1]] = dpk
        dn = ops.gemm(dp, self.proj_out_wT)
        dh, _ = ops.ln_modulate_bwd(dn, h_final, mod[:, D:], (B * hh * ww) // mod.shape[0])
        return dh

    @staticmethod
    def _key_bias(mask, B, Sk, dev):
        if mask is None:
            return None
        return ((1.0 - mask.to(device=dev, dtype=F32).reshape(B, Sk)) * -10000.0).contiguous()        # pixart/transformer.py:566-568

    @torch.no_grad()
    def _forward_trunk(self, latents, enc, mask, timestep, added_cond_kwargs):
        if not self._prepared:
            self.prepare()
        B, _, Hh, Ww = latents.shape
        hh, ww = Hh // 2, Ww // 2
        S, Sk = hh * ww, enc.shape[1]
        h = self._patch_embed(latents)
        t6, emb = self._conditioning(timestep, added_cond_kwargs, B, Hh, Ww)
        ctx2d = self._caption(enc)
        kb = self._key_bias(mask, B, Sk, self.device_)
        for blk in self.blocks:
            h, _ = self._block_fwd(blk, h, ctx2d, kb, t6, B, S, Sk, save=False)
        out, _ = self._head(h, emb, B, hh, ww)
        return out

    # ---- LoRA training of the trunk (pixart/model.py adapter path; the reference's published PixArt rows are adapter runs) ----
    def _engine_forward_lora(self, latents, enc, mask, timestep, added_cond_kwargs, save: bool):
        """the trunk forward keeping what the backward needs; TREAD routing (pixart/transformer.py:487-489 `set_router`, the routed span of the block loop) while
        training: between a route's two blocks every sample keeps a subset of its tokens (absolute position embeddings were added at the patch embedding, the
        AdaLN-single rows are per sample: nothing else to re-route)"""
        if not self._prepared:
            self.prepare()
        for g in self.lora_groups:
            g.pack()
        B, _, Hh, Ww = latents.shape
        hh, ww = Hh // 2, Ww // 2
        S, Sk, D = hh * ww, enc.shape[1], self.inner_dim
        h = self._patch_embed(latents)
        t6, emb = self._conditioning(timestep, added_cond_kwargs, B, Hh, Ww)
        if t6.shape[0] != B:
            raise NotImplementedError("PixArt LoRA training takes per-sample timesteps (tokenwise: the frozen forward only)")
        ctx2d = self._caption(enc)
        kb = self._key_bias(mask, B, Sk, self.device_)
        n = len(self.blocks)
        from ..training.tread import normalise_routes
        routes = normalise_routes(self._tread_routes, n) if (save and self.training and self._tread_router is not None) else []
        ctx = SimpleNamespace(B=B, S=S, Sk=Sk, hh=hh, ww=ww, ctx2d=ctx2d, kb=kb, blocks=[None] * n, S_of=[S] * n, route_start={}, route_end={})
        rp, info, saved, S_cur = 0, None, None, S
        for i, blk in enumerate(self.blocks):
            if rp < len(routes) and info is None and i == routes[rp]["start_layer_idx"]:
                info = self._tread_router.get_mask(h.view(B, S, D), mask_ratio=routes[rp]["selection_ratio"], force_keep=getattr(self, "_force_keep_mask", None))
                saved = h
                h = ops.gather_rows(h.view(B, S, D), info.keep_i32()).view(-1, D)                    # TREADRouter.start_route
                S_cur = info.ids_keep.shape[1]
                ctx.route_start[i] = info
            ctx.S_of[i] = S_cur
            h, ctx.blocks[i] = self._block_fwd(blk, h, ctx2d, kb, t6, B, S_cur, Sk, save)
            if info is not None and i == routes[rp]["end_layer_idx"]:
                full_seq = saved.clone()                                                             # TREADRouter.end_route(original_x=saved)
                ops.scatter_rows(h.view(B, S_cur, D), info.keep_i32(), full_seq.view(B, S, D))
                h, ctx.route_end[i] = full_seq, info
                info, saved, S_cur, rp = None, None, S, rp + 1
        if info is not None:
            raise ValueError("TREAD route does not end inside the block stack (end_layer_idx)")
        out, mod_out = self._head(h, emb, B, hh, ww)
        if save:
            ctx.h_final, ctx.mod_out = h, mod_out
        return out, ctx

    def _engine_backward_lora(self, ctx, dout):
        B, S, Sk, D = ctx.B, ctx.S, ctx.Sk, self.inner_dim
        dh = self._head_bwd(dout, ctx.h_final, ctx.mod_out, B, ctx.hh, ctx.ww)
        d_full = None
        for i in range(len(self.blocks) - 1, -1, -1):
            if i in ctx.route_end:                          # backward enters a route at its END: the routed blocks see only the kept tokens' gradient rows
                d_full = dh
                dh = ops.gather_rows(d_full.view(B, S, D), ctx.route_end[i].keep_i32()).view(-1, D)
            sv, ctx.blocks[i] = ctx.blocks[i], None
            dh = self._block_bwd(self.blocks[i], sv, dh, ctx.ctx2d, ctx.kb, B, ctx.S_of[i], Sk)
            if i in ctx.route_start:                        # ... and leaves it at its START: skipped tokens keep the gradient they had at the route's end
                ops.scatter_rows(dh.view(B, ctx.S_of[i], D), ctx.route_start[i].keep_i32(), d_full.view(B, S, D))
                dh, d_full = d_full, None
        return None

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, added_cond_kwargs=None, cross_attention_kwargs=None, attention_mask=None,
                encoder_attention_mask=None, return_dict: bool = True, force_keep_mask=None, **unsupported):
        self._force_keep_mask = force_keep_mask            # TREAD: tokens that may never be routed away
        for k, v in unsupported.items():
            if v is not None and v is not False:
                raise NotImplementedError(f"PixArtTransformer2DModel(st355): argument {k!r} is not supported on the HIP path")
        if attention_mask is not None or cross_attention_kwargs:
            raise NotImplementedError("PixArt(st355): self-attention masks / cross_attention_kwargs are not supported")
        if torch.is_grad_enabled() and self._lora_params:
            out = _PixArtLoraFn.apply(self, hidden_states, encoder_hidden_states, encoder_attention_mask, timestep, added_cond_kwargs, *self._lora_params)
        elif self._lora_params:
            with torch.no_grad():
                out, _ = self._engine_forward_lora(hidden_states, encoder_hidden_states, encoder_attention_mask, timestep, added_cond_kwargs, save=False)
        else:
            out = self._forward_trunk(hidden_states, encoder_hidden_states, encoder_attention_mask, timestep, added_cond_kwargs)
        return (out,) if not return_dict else SimpleNamespace(sample=out)


class _PixArtLoraFn(torch.autograd.Function):
    """one autograd node for the whole trunk under LoRA training (see flux/transformer.py::_FluxFn)"""

    @staticmethod
    def forward(fctx, model, latents, enc, mask, timestep, ack, *lora_params):
        out, ctx = model._engine_forward_lora(latents.detach(), enc.detach(), None if mask is None else mask.detach(), timestep.detach(), ack, True)
        fctx.model, fctx.ectx = model, ctx
        return out

    @staticmethod
    def backward(fctx, dout):
        model = fctx.model
        if model.grad_sync is not None:
            model.grad_sync.begin()
        model._engine_backward_lora(fctx.ectx, dout)
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
        return (None,) * 6 + tuple(grads)


class PixArtSigmaControlNetTransformerModel(CheckpointPlanMixin, nn.Module):
    """trunk (frozen) + ControlNet adapter (trainable): pixart/controlnet.py:166-326.  `num_layers` copied blocks; `from_transformer` semantics:
    the adapter blocks start as copies of trunk blocks 0..N-1, before/after projections start at zero."""

    def __init__(self, transformer: PixArtTransformer2DModel, num_layers: int = 13, init_from_transformer: bool = True):
        super().__init__()
        self.transformer = transformer
        self.blocks_num = num_layers
        self.config = transformer.config
        T = transformer
        D, dev = T.inner_dim, T.device_
        specs = []

        def alloc(name, shape):
            s = SimpleNamespace(name=name, shape=shape, numel=int(torch.tensor(shape).prod()))
            specs.append(s)
            return s

        self._cblocks_specs = []
        shells = []
        for i in range(num_layers):
            p = f"controlnet.controlnet_blocks.{i}."
            extra = {}
            if i == 0:
                extra["before_proj.weight"], extra["before_proj.bias"] = alloc(p + "before_proj.weight", (D, D)), alloc(p + "before_proj.bias", (D,))
            blk = _Block(self, p + "transformer_block.", D, T.H, T.hd, D, alloc, trainable=True)
            extra["after_proj.weight"], extra["after_proj.bias"] = alloc(p + "after_proj.weight", (D, D)), alloc(p + "after_proj.bias", (D,))
            shells.append((blk, extra))
        total = 0
        for s in specs:
            s.off = total
            total += (s.numel + 7) // 8 * 8
        self.arena = torch.zeros(total, dtype=BF16, device=dev)
        self.grad_arena = torch.zeros(total, dtype=BF16, device=dev)
        params = []
        for s in specs:
            s.t = self.arena[s.off:s.off + s.numel].view(*s.shape)
            s.g = self.grad_arena[s.off:s.off + s.numel].view(*s.shape)
            par = nn.Parameter(s.t, requires_grad=True)
            _attach(self, s.name, par)
            params.append(par)
        self._params = params
        self._offsets = [(s.off, s.numel) for s in specs]
        self.cblocks = []
        for blk, extra in shells:
            blk.G = {k: v.g for k, v in blk.P.items()}
            blk.P = {k: v.t for k, v in blk.P.items()}
            ex = SimpleNamespace(**{k.replace(".", "_"): v.t for k, v in extra.items()})
            ex.G = {k: v.g for k, v in extra.items()}
            self.cblocks.append((blk, ex))
        if init_from_transformer:
            with torch.no_grad():
                for i, (blk, _) in enumerate(self.cblocks):
                    for k, v in blk.P.items():
                        v.copy_(T.blocks[i].P[k])
        self.full = True
        self.grad_sync = None
        self._last_grad_flat = None

    def trainable_parameters(self):
        return list(self._params)

    @torch.no_grad()
    def init_adapter_synthetic(self, seed: int = 7, std: float = 0.02):
        """non-zero before/after projections (true zero-init would give zero gradients everywhere but after_proj on the first step)"""
        g = torch.Generator(device=self.transformer.device_).manual_seed(seed)
        for _, ex in self.cblocks:
            for k in vars(ex):
                if k.endswith("proj_weight"):
                    getattr(ex, k).copy_(torch.randn(getattr(ex, k).shape, generator=g, device=self.transformer.device_) * std)

    def adapter_state_dict(self) -> Dict[str, torch.Tensor]:
        """adapter weights under the reference's names (`controlnet_blocks.{i}.before_proj|transformer_block.*|after_proj`)"""
        sd = {}
        for i, (blk, ex) in enumerate(self.cblocks):
            p = f"controlnet_blocks.{i}."
            for k, v in blk.P.items():
                sd[p + "transformer_block." + k] = v.detach().clone()
            for k in ("before_proj", "after_proj"):
                if hasattr(ex, k + "_weight"):
                    sd[p + k + ".weight"], sd[p + k + ".bias"] = getattr(ex, k + "_weight").detach().clone(), getattr(ex, k + "_bias").detach().clone()
        return sd

    def _engine_forward(self, latents, cond, enc, mask, timestep, added_cond_kwargs, save: bool):
        T = self.transformer
        if not T._prepared:
            T.prepare()
        dev = T.device_
        B, _, Hh, Ww = latents.shape
        hh, ww = Hh // 2, Ww // 2
        S, Sk = hh * ww, enc.shape[1]
        for blk, _ in self.cblocks:
            blk.refresh(dev)                                   # the adapter's padded working copies follow its parameters
        h = T._patch_embed(latents)
        cs = T._patch_embed(cond)
        if timestep.dim() != 1 and timestep.numel() != 1:
            raise NotImplementedError("PixArt ControlNet wrapper: per-sample timesteps only (the reference's wrapper calls adaln_single directly, pixart/controlnet.py:241-246)")
        t6, emb = T._conditioning(timestep, added_cond_kwargs, B, Hh, Ww)
        ctx2d = T._caption(enc)
        kb = T._key_bias(mask, B, Sk, dev)
        ctx = SimpleNamespace(B=B, S=S, Sk=Sk, hh=hh, ww=ww, ctx2d=ctx2d, kb=kb, t6=t6, trunk={}, ctrl={}, cs_in={}, cs0=cs, ck={})
        # activation-checkpoint plan over the UNITS of the wrapper's loop (unit i = control block i-1, if any, + trunk block i).  The reference's wrapper leaves
        # checkpointing as a TODO (pixart/controlnet.py:270-273) and its trunk plans per block (pixart/transformer.py:627-700); same planner here
        ctx.segs = self._checkpoint_segments(len(T.blocks)) if save else [(i, 1, False) for i in range(len(T.blocks))]
        for (s0, n, ck) in ctx.segs:
            if ck:
                ctx.ck[s0] = (h, cs)
            for i in range(s0, s0 + n):
                h, cs = self._unit_fwd(i, h, cs, ctx, save and not ck)
        out, mod_out = T._head(h, emb, B, hh, ww)
        if save:
            ctx.h_final, ctx.mod_out = h, mod_out
        return out, ctx

    def _unit_fwd(self, i: int, h, cs, ctx, save: bool):
        """unit i of the wrapper's loop (pixart/controlnet.py:266-297): control block i-1 into the trunk's residual stream, then trunk block i"""
        T = self.transformer
        B, S, Sk, ctx2d, kb, t6 = ctx.B, ctx.S, ctx.Sk, ctx.ctx2d, ctx.kb, ctx.t6
        if 0 < i <= self.blocks_num:
            blk, ex = self.cblocks[i - 1]
            if i == 1:
                cs = ops.gemm(cs, ex.before_proj_weight, bias=ex.before_proj_bias, epilogue=EPI_ADD, aux_in=h)
            cs, svc = T._block_fwd(blk, cs, ctx2d, kb, t6, B, S, Sk, save, exact=True)
            h = ops.gemm(cs, ex.after_proj_weight, bias=ex.after_proj_bias, epilogue=EPI_ADD, aux_in=h)
            if save:
                ctx.ctrl[i - 1], ctx.cs_in[i - 1] = svc, cs
        h, sv = T._block_fwd(T.blocks[i], h, ctx2d, kb, t6, B, S, Sk, save and i >= 1, exact=True)
        if save:
            ctx.trunk[i] = sv
        return h, cs

    def _engine_backward(self, ctx, dout):
        T = self.transformer
        B, S, Sk = ctx.B, ctx.S, ctx.Sk
        dh = T._head_bwd(dout, ctx.h_final, ctx.mod_out, B, ctx.hh, ctx.ww)
        dcs = None
        for i in range(len(T.blocks) - 1, 0, -1):                  # trunk block 0 has nothing trainable upstream of it
            if i not in ctx.trunk:                                   # a checkpointed segment: re-run it from its kept input, keeping the activations this time
                s0, n = next((a, c) for (a, c, ck) in ctx.segs if ck and a <= i < a + c)
                hr, cr = ctx.ck.pop(s0)
                for u in range(s0, s0 + n):
                    hr, cr = self._unit_fwd(u, hr, cr, ctx, True)
                del hr, cr
            dh = T._block_bwd(T.blocks[i], ctx.trunk.pop(i), dh, ctx.ctx2d, ctx.kb, B, S, Sk)
            if i <= self.blocks_num:
                blk, ex = self.cblocks[i - 1]
                cs_out = ctx.cs_in.pop(i - 1)
                # h' = h + after_proj(cs_out):  d after_proj, d cs_out (+ what the next control block sent back)
                ops.gemm_tn(_p64(dh), _p64(cs_out), out=ex.G["after_proj.weight"])
                tb = torch.empty(1, dh.shape[1], dtype=F32, device=dh.device)
                ops.colsum_prod(dh, tb); ex.G["after_proj.bias"].copy_(tb[0])
                wT = ex.after_proj_weight.t().contiguous()
                dcs = ops.gemm(dh, wT) if dcs is None else ops.gemm(dh, wT, epilogue=EPI_ADD, aux_in=dcs)
                dcs = T._block_bwd(blk, ctx.ctrl.pop(i - 1), dcs, ctx.ctx2d, ctx.kb, B, S, Sk)
                if i == 1:                                         # cs_in = h + before_proj(cs0): h gets dcs too (unused: nothing trainable before it)
                    ops.gemm_tn(_p64(dcs), _p64(ctx.cs0), out=ex.G["before_proj.weight"])
                    ops.colsum_prod(dcs, tb); ex.G["before_proj.bias"].copy_(tb[0])

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, controlnet_cond=None, added_cond_kwargs=None, cross_attention_kwargs=None,
                attention_mask=None, encoder_attention_mask=None, return_dict: bool = True):
        if attention_mask is not None or cross_attention_kwargs:
            raise NotImplementedError("PixArt ControlNet(st355): self-attention masks / cross_attention_kwargs are not supported")
        if controlnet_cond is None:
            out = self.transformer._forward_trunk(hidden_states, encoder_hidden_states, encoder_attention_mask, timestep, added_cond_kwargs)
        elif torch.is_grad_enabled():
            out = _CtrlFn.apply(self, hidden_states, controlnet_cond, encoder_hidden_states, encoder_attention_mask, timestep, added_cond_kwargs, *self._params)
        else:
            with torch.no_grad():
                out, _ = self._engine_forward(hidden_states, controlnet_cond, encoder_hidden_states, encoder_attention_mask, timestep, added_cond_kwargs, False)
        return (out,) if not return_dict else SimpleNamespace(sample=out)


class _CtrlFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, model, latents, cond, enc, mask, timestep, ack, *params):
        out, ctx = model._engine_forward(latents.detach(), cond.detach(), enc.detach(), None if mask is None else mask.detach(), timestep.detach(), ack, True)
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
            model.grad_sync.ready(0, model.grad_arena.numel())
            model.grad_scale_from_sync = model.grad_sync.finish()
        from ..training.grad_sync import hand_over_gradients
        gflat = hand_over_gradients(model, model.grad_arena)
        model._last_grad_flat = gflat
        grads = [gflat[off:off + n].view_as(p) for (off, n), p in zip(model._offsets, model._params)]
        return (None,) * 7 + tuple(grads)
