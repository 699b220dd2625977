"""Tensor-level wrappers over the C ABI (include/st355.h).

torch is used here only as the owner of device memory and of the HIP stream: every wrapper validates
shapes/dtypes, allocates the outputs, and passes raw device pointers + the current stream to libst355.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import lib as _l
from .lib import EPI_ADD, EPI_GATE_RESIDUAL, EPI_GEGLU, EPI_GEGLU_GRAD, EPI_GELU, EPI_HEADS, EPI_MUL_GELU_GRAD, EPI_NONE, EPI_QK_NORM_ROPE, GemmArgs, Heads, QkRope  # noqa: F401

BF16 = torch.bfloat16
F32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _l.St355Error(f"{name}: expected a device tensor (the train step has no CPU path)")
    if t.dtype != dtype:
        raise _l.St355Error(f"{name}: expected dtype {dtype}, got {t.dtype}")


def _dev(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _l.St355Error(f"{name}: expected a device tensor (the train step has no CPU path)")


def _rows(t: torch.Tensor, name: str) -> int:
    """leading dimension (elements) of a 2-D row-major view"""
    if t.dim() != 2 or t.stride(1) != 1:
        raise _l.St355Error(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} stride {t.stride()}")
    return t.stride(0)


def _seg(t: torch.Tensor, name: str):
    """A GEMM / skinny operand is either a 2-D row-major view or a 3-D view [segments, rows, cols] with unit inner stride whose segment stride is a
    whole number of rows: the per-sample row blocks of a joint [B, S, *] buffer, e.g. `qkv.view(B, S, 3 * D)[:, St:]` (st355_gemm_args.seg_rows).
    Returns (rows_total, cols, ld, seg_rows, seg_stride_rows); seg_rows = 0 for a plain 2-D operand."""
    if t.dim() == 2:
        return t.shape[0], t.shape[1], _rows(t, name), 0, 0
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) <= 0 or t.stride(0) % t.stride(1) != 0:
        raise _l.St355Error(f"{name}: expected a 2-D row-major view or a 3-D [segments, rows, cols] view whose segment stride is a whole number of rows, "
                            f"got shape {tuple(t.shape)} stride {t.stride()}")
    return t.shape[0] * t.shape[1], t.shape[2], t.stride(1), t.shape[1], t.stride(0) // t.stride(1)


def _seg_join(cur: int, new: int, name: str) -> int:
    if new and cur and new != cur:
        raise _l.St355Error(f"{name}: segmented operands of one problem must share seg_rows ({new} vs {cur})")
    return cur or new


# ------------------------------------------------------------------------------------------------
# streaming ops
# ------------------------------------------------------------------------------------------------
def flow_noise_mix(x, sigma, noise=None, seed: int = 0, offset: int = 0, want_target: bool = True):
    """x_t = (1-σ)x + σn ; target = n - x   (common.py:4975-4992, 4610-4611).  Returns (x_t, target, noise)."""
    L = _l.load()
    _chk(x, BF16, "x"); _chk(sigma, F32, "sigma")
    x = x.contiguous()
    B = x.shape[0]
    per = x.numel() // B
    x_t = torch.empty_like(x)
    target = torch.empty_like(x) if want_target else None
    if noise is not None:
        _chk(noise, BF16, "noise")
        noise = noise.contiguous()
        noise_out = None
    else:
        noise_out = torch.empty_like(x)
    _l.check(L.st355_flow_noise_mix(_stream(), _ptr(x), _ptr(noise), _ptr(sigma), _ptr(x_t), _ptr(target), _ptr(noise_out),
                                    B, per, seed, offset), "flow_noise_mix")
    return x_t, target, (noise if noise is not None else noise_out)


def ddpm_noise_mix(x, noise, sqrt_acp, sqrt_1macp, want_v: bool = True):
    L = _l.load()
    _chk(x, BF16, "x"); _chk(noise, BF16, "noise"); _chk(sqrt_acp, F32, "sqrt_acp"); _chk(sqrt_1macp, F32, "sqrt_1macp")
    x = x.contiguous(); noise = noise.contiguous()
    B = x.shape[0]
    x_t = torch.empty_like(x)
    v = torch.empty_like(x) if want_v else None
    _l.check(L.st355_ddpm_noise_mix(_stream(), _ptr(x), _ptr(noise), _ptr(sqrt_acp), _ptr(sqrt_1macp), _ptr(x_t), _ptr(v),
                                    B, x.numel() // B), "ddpm_noise_mix")
    return x_t, v


def mse_loss(pred, target, weight=None, want_grad: bool = True, grad_scale: float = 1.0):
    """mean_b(mean_chw((pred-target)^2 * w_b)) in fp32 + fused d(loss)/d(pred).  Returns (loss[1], per_sample[B], dpred)."""
    L = _l.load()
    _chk(pred, BF16, "pred"); _chk(target, BF16, "target")
    pred = pred.contiguous(); target = target.contiguous()
    B = pred.shape[0]
    loss = torch.empty(1, dtype=F32, device=pred.device)
    per_sample = torch.empty(B, dtype=F32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    if weight is not None:
        _chk(weight, F32, "weight")
    _l.check(L.st355_mse_loss(_stream(), _ptr(pred), _ptr(target), _ptr(weight), _ptr(loss), _ptr(per_sample), _ptr(dpred),
                              B, pred.numel() // B, grad_scale), "mse_loss")
    return loss, per_sample, dpred


LOSS_TYPES = {"l2": 0, "huber": 1, "smooth_l1": 2}


def cond_loss(pred, target, loss_type: str = "l2", huber_c=0.1, weight=None, want_grad: bool = True, grad_scale: float = 1.0, emask=None):
    """conditional_loss(reduction='none') -> [* emask] -> per-sample mean -> batch mean (common.py:6132-6166, 6402-6429) + fused d(loss)/d(pred).
    huber_c: float or fp32 device tensor [B] (scheduled huber).  emask: fp32 [B, H*W] element mask broadcast over the channels (conditioning-mask
    losses) or None.  Returns (loss[1], per_sample[B], dpred)."""
    L = _l.load()
    _chk(pred, BF16, "pred"); _chk(target, BF16, "target")
    pred = pred.contiguous(); target = target.contiguous()
    B = pred.shape[0]
    loss = torch.empty(1, dtype=F32, device=pred.device)
    per_sample = torch.empty(B, dtype=F32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    if loss_type != "l2":
        if not torch.is_tensor(huber_c):
            huber_c = torch.full((B,), float(huber_c), dtype=F32, device=pred.device)
        _chk(huber_c, F32, "huber_c")
        huber_c = huber_c.contiguous()
    else:
        huber_c = None
    if weight is not None:
        _chk(weight, F32, "weight")
    period = 0
    if emask is not None:
        _chk(emask, F32, "emask")
        emask = emask.reshape(B, -1).contiguous()
        period = emask.shape[1]
    _l.check(L.st355_cond_loss_masked(_stream(), _ptr(pred), _ptr(target), _ptr(weight), _ptr(huber_c), LOSS_TYPES[loss_type], _ptr(emask), period,
                                      _ptr(loss), _ptr(per_sample), _ptr(dpred), B, pred.numel() // B, grad_scale), "cond_loss")
    return loss, per_sample, dpred


def flux_pack(latents):
    L = _l.load()
    _chk(latents, BF16, "latents")
    latents = latents.contiguous()
    B, Cc, H, W = latents.shape
    out = torch.empty(B, (H // 2) * (W // 2), Cc * 4, dtype=BF16, device=latents.device)
    _l.check(L.st355_flux_pack(_stream(), _ptr(latents), _ptr(out), B, Cc, H, W), "flux_pack")
    return out


def flux_unpack(packed, Cc: int, H: int, W: int):
    L = _l.load()
    _chk(packed, BF16, "packed")
    packed = packed.contiguous()
    B = packed.shape[0]
    out = torch.empty(B, Cc, H, W, dtype=BF16, device=packed.device)
    _l.check(L.st355_flux_unpack(_stream(), _ptr(packed), _ptr(out), B, Cc, H, W), "flux_unpack")
    return out


def patchify(latents, order: int = 0):
    """[B,C,H,W] -> [B,(H/2)(W/2),4C]; order 0 = (c,dh,dw) features (Flux pack / PatchEmbed im2col), 1 = (dh,dw,c) (SD3/PixArt)"""
    L = _l.load()
    _chk(latents, BF16, "latents")
    latents = latents.contiguous()
    B, Cc, H, W = latents.shape
    out = torch.empty(B, (H // 2) * (W // 2), Cc * 4, dtype=BF16, device=latents.device)
    _l.check(L.st355_patchify(_stream(), _ptr(latents), _ptr(out), B, Cc, H, W, order), "patchify")
    return out


def unpatchify(packed, Cc: int, H: int, W: int, order: int = 0):
    L = _l.load()
    _chk(packed, BF16, "packed")
    packed = packed.contiguous()
    B = packed.shape[0]
    out = torch.empty(B, Cc, H, W, dtype=BF16, device=packed.device)
    _l.check(L.st355_unpatchify(_stream(), _ptr(packed), _ptr(out), B, Cc, H, W, order), "unpatchify")
    return out


def timestep_proj(t, dim: int, scale: float = 1.0):
    L = _l.load()
    _chk(t, F32, "t")
    out = torch.empty(t.shape[0], dim, dtype=BF16, device=t.device)
    _l.check(L.st355_timestep_proj(_stream(), _ptr(t.contiguous()), _ptr(out), t.shape[0], dim, scale), "timestep_proj")
    return out


def silu(x):
    L = _l.load()
    _chk(x, BF16, "x")
    x = x.contiguous()
    y = torch.empty_like(x)
    _l.check(L.st355_silu(_stream(), _ptr(x), _ptr(y), x.numel()), "silu")
    return y


def gelu_tanh(x):
    L = _l.load()
    _chk(x, BF16, "x")
    x = x.contiguous()
    y = torch.empty_like(x)
    _l.check(L.st355_gelu_tanh(_stream(), _ptr(x), _ptr(y), x.numel()), "gelu_tanh")
    return y


def silu_bwd(x, dy):
    """dx = dy * silu'(x)"""
    L = _l.load()
    _chk(x, BF16, "x"); _chk(dy, BF16, "dy")
    x = x.contiguous(); dy = dy.contiguous()
    dx = torch.empty_like(x)
    _l.check(L.st355_silu_bwd(_stream(), _ptr(x), _ptr(dy), _ptr(dx), x.numel()), "silu_bwd")
    return dx


def add(a, b):
    L = _l.load()
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    a = a.contiguous(); b = b.contiguous()
    y = torch.empty_like(a)
    _l.check(L.st355_add(_stream(), _ptr(a), _ptr(b), _ptr(y), a.numel()), "add")
    return y


def gather_rows(x, idx, out=None):
    """out[b, j, :] = x[b, idx[b, j], :]; x [B, S, D] bf16 (last dim contiguous), idx [B, K] int32 -> [B, K, D]  (TREADRouter.start_route)"""
    L = _l.load()
    _dev(x, "x"); _dev(idx, "idx")
    if x.dtype != BF16 or x.dim() != 3 or x.stride(2) != 1 or idx.dtype != torch.int32 or not idx.is_contiguous():
        raise _l.St355Error("gather_rows: x [B, S, D] bf16 with contiguous rows, idx [B, K] contiguous int32")
    B, S, D = x.shape
    K = idx.shape[1]
    if out is None:
        out = torch.empty(B, K, D, dtype=BF16, device=x.device)
    _l.check(L.st355_gather_rows(_stream(), _ptr(x), x.stride(1), x.stride(0), _ptr(idx), _ptr(out), out.stride(1), out.stride(0), B, K, D), "gather_rows")
    return out


def scatter_rows(src, idx, dst):
    """dst[b, idx[b, j], :] = src[b, j, :] in place; src [B, K, D], dst [B, S, D] bf16, idx [B, K] int32  (TREADRouter.end_route)"""
    L = _l.load()
    _dev(src, "src"); _dev(dst, "dst"); _dev(idx, "idx")
    if src.dtype != BF16 or dst.dtype != BF16 or src.dim() != 3 or dst.dim() != 3 or src.stride(2) != 1 or dst.stride(2) != 1 or idx.dtype != torch.int32:
        raise _l.St355Error("scatter_rows: src [B, K, D] / dst [B, S, D] bf16 with contiguous rows, idx [B, K] int32")
    B, K, D = src.shape
    _l.check(L.st355_scatter_rows(_stream(), _ptr(src), src.stride(1), src.stride(0), _ptr(idx.contiguous()), _ptr(dst), dst.stride(1), dst.stride(0), B, K, D),
             "scatter_rows")
    return dst


def scale_cols(x, gate, rows_per_batch: int, out=None):
    """out[m,n] = x[m,n] * gate[m // rows_per_batch, n]"""
    L = _l.load()
    _chk(x, BF16, "x"); _chk(gate, BF16, "gate")
    ldx = _rows(x, "x"); gs = _rows(gate, "gate")
    M, N = x.shape
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=x.device)
    _l.check(L.st355_scale_cols(_stream(), _ptr(x), ldx, _ptr(gate), gs, rows_per_batch, _ptr(out), _rows(out, "out"), M, N),
             "scale_cols")
    return out


# ------------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------------
_gemm_ws = {}
_GEMM_WS_BYTES = 512 << 20           # ONE size for every caller (r5 advice: a smaller first request made a later one re-allocate under captured graphs): the largest
                                     # user is the weight-gradient split-K (7 slices of a 6144 x 1536 gradient: 264 MB); the stream-K tail keeps up to 128 x 3 tile slabs (96 MiB)
_gemm_flags = {}


def _gemm_tile_flags(dev):
    """st355_gemm_args.tile_flags: 1024 arrival counters of the stream-K tail, zero once — every launch leaves them zero"""
    f = _gemm_flags.get(dev.index)
    if f is None:
        if torch.cuda.is_current_stream_capturing():
            raise _l.St355Error("the GEMM tile counters would be allocated inside a hipGraph capture: run one eager step first")
        f = torch.zeros(1024, dtype=torch.int32, device=dev)
        _gemm_flags[dev.index] = f
    return f



_ws_retired = []       # outgrown scratch buffers stay alive: a captured hipGraph may still hold their addresses


def _gemm_workspace(dev, nbytes: int = _GEMM_WS_BYTES):
    """caller-owned fp32 scratch for the split-K paths (thin GEMMs, weight gradients); libst355 never allocates.  Grow-only; an outgrown buffer is retired, never
    freed (a hipGraph captured earlier keeps its address), and growing DURING a capture is refused (the capture would bake in a pointer of its private pool)"""
    ws = _gemm_ws.get(dev.index)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            if torch.cuda.is_current_stream_capturing():
                raise _l.St355Error(f"the shared GEMM scratch would have to grow ({ws.numel() * 4} -> {nbytes} bytes) inside a hipGraph capture: run one eager step first")
            _ws_retired.append(ws)
        ws = torch.empty(nbytes // 4, dtype=F32, device=dev)
        _gemm_ws[dev.index] = ws
    return ws


def qk_rope(Q, K, rrms, wq, wk, cos, sin, H: int, S: int, pos0: int, eps: float = 1e-6, Vt=None):
    """operands of the fused QKV projection epilogue (EPI_QK_NORM_ROPE; st355_qk_rope in st355.h): pass as gemm(..., rope=qk_rope(...)).
    Q, K: [B,H,S,128] bf16 outputs; rrms: [B*S, 2H] fp32 output; wq / wk: RMSNorm weights [128] or None; cos / sin: [S,64] fp32, one angle per
    interleaved channel pair (= full_table[:, 0::2])."""
    if cos.shape != (S, 64) or sin.shape != (S, 64) or not cos.is_contiguous() or not sin.is_contiguous():
        raise _l.St355Error(f"qk_rope: cos / sin must be contiguous [S={S}, 64] per-pair tables, got {tuple(cos.shape)}")
    _chk(Q, BF16, "Q"); _chk(K, BF16, "K"); _chk(rrms, F32, "rrms"); _chk(cos, F32, "cos"); _chk(sin, F32, "sin")
    r = QkRope()
    r.Q, r.K, r.rrms, r.wq, r.wk, r.cos, r.sin = _ptr(Q), _ptr(K), _ptr(rrms), _ptr(wq), _ptr(wk), _ptr(cos), _ptr(sin)
    r.H, r.S, r.pos0, r.eps = H, S, pos0, eps
    if Vt is not None:                  # also emit the head-major V^T [B,H,128,Sp] for attn_fwd (else: row-major V only, for attn_fwd_vrows)
        _chk(Vt, BF16, "Vt")
        r.Vt, r.Sp = _ptr(Vt), Vt.shape[-1]
    r._keep = (Q, K, rrms, wq, wk, cos, sin, Vt)
    return r


def heads(Q, K, Vt, H: int, S: int, pos0: int, n_q: int, n_k: int):
    """destinations of the head-splitting projection epilogue (EPI_HEADS; st355_heads in st355.h): pass as gemm(..., epilogue=EPI_HEADS, heads=heads(...),
    rows_per_batch=rows of this stream per sample).  Q / K: [B, H, S, 64] bf16 (None when the projection has no such part); Vt: [B, H, 64, Sp] or None."""
    h = Heads()
    for t, nm in ((Q, "Q"), (K, "K"), (Vt, "Vt")):
        if t is not None:
            _chk(t, BF16, nm)
            if not t.is_contiguous():
                raise _l.St355Error(f"heads: {nm} must be contiguous")
    h.Q, h.K, h.Vt = _ptr(Q), _ptr(K), _ptr(Vt)
    h.H, h.S, h.pos0, h.Sp, h.n_q, h.n_k = H, S, pos0, (Vt.shape[-1] if Vt is not None else 0), n_q, n_k
    h._keep = (Q, K, Vt)
    return h


def _gemm_args(g, a, w, bias=None, out=None, epilogue: int = EPI_NONE, a2=None, b2=None, aux_out=None, aux_in=None,
               gate=None, rows_per_batch: int = 0, k2_real: int = 0, rope=None, heads=None):
    """a, a2, out, aux_in, aux_out may be 3-D [segments, rows, cols] strided views (see _seg): one problem over the row blocks of a joint buffer.
    epilogue=EPI_QK_NORM_ROPE (rope=qk_rope(...), rows_per_batch=rows of this stream per sample): `out` is the V destination [M, N/3] (q / k go
    head-major to rope.Q / rope.K)."""
    _chk(a, BF16, "a"); _chk(w, BF16, "w")
    M, K, g.lda, seg, g.seg_a = _seg(a, "a")
    N, Kw = w.shape
    if K != Kw:
        raise _l.St355Error(f"gemm: K mismatch {K} vs {Kw}")
    if epilogue == EPI_QK_NORM_ROPE:
        if rope is None or out is None:
            raise _l.St355Error("gemm: EPI_QK_NORM_ROPE needs rope=qk_rope(...) and out= (the V destination)")
        g.rope = C.pointer(rope)
        g._rope_keep = rope
        g.rows_per_batch = rows_per_batch
    # output width: N, except the fused-QKV form (the V third), EPI_HEADS (the v heads), EPI_GEGLU (value * gelu(gate): N / 2) and EPI_GEGLU_GRAD (d value | d gate: 2 N)
    n_out = N // 3 if epilogue == EPI_QK_NORM_ROPE else N // 2 if epilogue == EPI_GEGLU else 2 * N if epilogue == EPI_GEGLU_GRAD else N
    if epilogue == EPI_HEADS:
        if heads is None:
            raise _l.St355Error("gemm: EPI_HEADS needs heads=heads(...)")
        n_out = N - heads.n_q - heads.n_k
        g.heads = C.pointer(heads)
        g._heads_keep = heads
        g.rows_per_batch = rows_per_batch
        if n_out == 0:                     # no v part (a q-only / k-only projection): C is never written — an [M, 8] scratch satisfies the argument checks
            if out is not None:
                raise _l.St355Error("gemm: EPI_HEADS without v heads takes no out=")
            n_out = 8
    if epilogue != EPI_QK_NORM_ROPE and out is None:
        out = torch.empty(M, n_out, dtype=BF16, device=a.device)
    Mo, No, g.ldc, sr, g.seg_c = _seg(out, "out")
    if (Mo, No) != (M, n_out):
        raise _l.St355Error(f"gemm: out is {Mo}x{No}, expected {M}x{n_out}")
    seg = _seg_join(seg, sr, "gemm")
    g.A, g.B, g.ldb, g.C = _ptr(a), _ptr(w), _rows(w, "w"), _ptr(out)
    g.M, g.N, g.K, g.K2 = M, N, K, 0
    if a2 is not None:
        _chk(a2, BF16, "a2"); _chk(b2, BF16, "b2")
        M2, K2, g.lda2, sr, g.seg_a2 = _seg(a2, "a2")
        seg = _seg_join(seg, sr, "gemm")
        if M2 != M or b2.shape[0] != N or K2 != b2.shape[1]:
            raise _l.St355Error("gemm: low-rank extension shape mismatch")
        g.A2 = _ptr(a2)
        g.B2, g.ldb2 = _ptr(b2), _rows(b2, "b2")
        g.K2 = K2
        g.K2_real = int(k2_real)          # adapter columns inside the 64-column granule (profiler accounting only)
    if bias is not None:
        _chk(bias, BF16, "bias")
        if not bias.is_contiguous():
            raise _l.St355Error("gemm: bias must be contiguous")
        g.bias = _ptr(bias)
    g.epilogue = epilogue
    if (N <= 128 and M >= 1024) or (M * N >= 128 * 65536 and epilogue <= EPI_ADD):       # thin problems (split-K slabs); 128+ tiles of 256x256 (stream-K tail)
        ws = _gemm_workspace(a.device)
        g.workspace, g.workspace_bytes = _ptr(ws), ws.numel() * 4
        g.tile_flags = _ptr(_gemm_tile_flags(a.device))
    if aux_out is not None:
        _chk(aux_out, BF16, "aux_out")
        _, _, g.ld_aux_out, sr, g.seg_out = _seg(aux_out, "aux_out")
        seg = _seg_join(seg, sr, "gemm")
        g.aux_out = _ptr(aux_out)
    if aux_in is not None:
        _chk(aux_in, BF16, "aux_in")
        _, _, g.ld_aux_in, sr, g.seg_in = _seg(aux_in, "aux_in")
        seg = _seg_join(seg, sr, "gemm")
        g.aux_in = _ptr(aux_in)
    if gate is not None:
        _chk(gate, BF16, "gate")
        g.gate, g.gate_stride = _ptr(gate), _rows(gate, "gate")
        g.rows_per_batch = rows_per_batch
    g.seg_rows = seg
    return out


def gemm(a, w, **kw):
    """out[M,N] = a[M,K] @ w[N,K]^T (+ a2[M,K2] @ b2[N,K2]^T) with a fused epilogue (see st355.h).
    kwargs: bias, out, epilogue, a2, b2, aux_out, aux_in, gate, rows_per_batch."""
    L = _l.load()
    g = GemmArgs()
    out = _gemm_args(g, a, w, **kw)
    _l.check(L.st355_gemm_bf16(_stream(), C.byref(g)), "gemm_bf16")
    return out


def sum_chunks_bf16(chunks, world: int, out):
    """out[i] = bf16(sum_w float(chunks[w * n + i])), n = out.numel(): the fp32-accumulating local half of the all-to-all reduce-scatter (grad_sync)"""
    L = _l.load()
    _chk(chunks, BF16, "chunks"); _chk(out, BF16, "out")
    n = out.numel()
    if chunks.numel() != world * n:
        raise _l.St355Error(f"sum_chunks_bf16: chunks holds {chunks.numel()} elements, expected world * n = {world * n}")
    _l.check(L.st355_sum_chunks_bf16(_stream(), _ptr(chunks), int(world), int(n), _ptr(out)), "sum_chunks_bf16")
    return out


def geglu_interleave(w, bias=None):
    """the feed-forward projection's rows (nn.Linear.weight [2F, K] = [value rows | gate rows]) in the order the EPI_GEGLU / EPI_GEGLU_GRAD epilogues contract
    them: every 64 rows = 32 value features followed by the 32 gate features of the SAME indices (st355.h).  Returns (w_il, bias_il); built once for frozen weights."""
    N2, K = w.shape
    F_ = N2 // 2
    if N2 % 64 or F_ % 32:
        raise _l.St355Error(f"geglu_interleave: 2F = {N2} must be a multiple of 64")
    wi = torch.stack([w[:F_].view(F_ // 32, 32, K), w[F_:].view(F_ // 32, 32, K)], dim=1).reshape(N2, K).contiguous()
    bi = None if bias is None else torch.stack([bias[:F_].view(F_ // 32, 32), bias[F_:].view(F_ // 32, 32)], dim=1).reshape(N2).contiguous()
    return wi, bi


def gemm_set_tail_split(mode: int) -> int:
    """st355_gemm_set_tail_split: 1 = stream-K tail where it applies (default), 0 = the uncut schedules, -1 = environment; returns the previous mode"""
    return int(_l.load().st355_gemm_set_tail_split(int(mode)))


def gemm_tail_placement() -> int:
    """st355_gemm_tail_placement: 0 = the XCD placement probe has not run on this device, 1 = round-robin confirmed (stream-K tail in use), 2 = not confirmed"""
    return int(_l.load().st355_gemm_tail_placement())


def gemm_set_persistent(mode: int) -> int:
    """st355_gemm_set_persistent: 1 = persistent tile walk (k_gemm_pz) where it applies, 0 = one tile per workgroup, -1 = default; returns the previous mode"""
    return int(_l.load().st355_gemm_set_persistent(int(mode)))


def attn_set_impl(fwd: int = -1, dq: int = -1, dkv: int = -1):
    """st355_attn_set_impl: fwd / dq 64 = the hand-scheduled 64-rows-per-wave kernels where they apply (default), 32 = the 32-row kernels everywhere;
    dkv 4 = the hand-scheduled dK/dV body (default), 3 = k_attn_bwd_dkv3; -1 = unchanged.  Returns the previous (fwd, dq, dkv)."""
    prev = int(_l.load().st355_attn_set_impl(int(fwd), int(dq), int(dkv)))
    return prev // 65536, (prev // 256) % 256, prev % 256


def gemm_grouped(problems):
    """run several independent GEMMs that share one epilogue kind in as few launches as possible.
    problems: list of dicts with keys a, w and the kwargs of gemm().  Returns the list of outputs."""
    L = _l.load()
    arr = (GemmArgs * len(problems))()
    outs = []
    for g, pr in zip(arr, problems):
        pr = dict(pr)
        outs.append(_gemm_args(g, pr.pop("a"), pr.pop("w"), **pr))
    _l.check(L.st355_gemm_bf16_grouped(_stream(), arr, len(problems)), "gemm_bf16_grouped")
    return outs


def fp8_quantize_weight(w):
    """quantize_weight_to_fp8 (fp8_native.py:25-30): returns (q uint8 [N,K] holding e4m3fn bytes, scale fp32 [N])"""
    L = _l.load()
    _chk(w, BF16, "w")
    N, K = w.shape
    q = torch.empty(N, K, dtype=torch.uint8, device=w.device)
    scale = torch.empty(N, dtype=F32, device=w.device)
    _l.check(L.st355_fp8_quantize_weight(_stream(), _ptr(w), _rows(w, "w"), _ptr(q), _ptr(scale), N, K), "fp8_quantize_weight")
    return q, scale


def fp8_quantize_act(x):
    """per-call e5m2 quantisation of the activations (fp8_native.py:58-60): returns (q uint8 [M,K], scale_a fp32 [1] = 1/input_scale)"""
    L = _l.load()
    _chk(x, BF16, "x")
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    scale_a = torch.empty(1, dtype=F32, device=x.device)
    ws = torch.empty(1, dtype=torch.int32, device=x.device)
    _l.check(L.st355_fp8_quantize_act(_stream(), _ptr(x), _rows(x, "x"), _ptr(q), _ptr(scale_a), M, K, _ptr(ws)), "fp8_quantize_act")
    return q, scale_a


def linear_fp8(xq, scale_a, wq, w_scale, bias=None, out=None):
    """out = (xq wq^T) * scale_a * w_scale[n] + bias  -> bf16   (torch._scaled_mm with row-wise scales, fp8_native.py:64-75)"""
    L = _l.load()
    for t, nm in ((xq, "xq"), (wq, "wq")):
        _dev(t, nm)
        if t.dtype != torch.uint8 or not t.is_contiguous():
            raise _l.St355Error(f"linear_fp8: {nm} must be a contiguous uint8 (fp8 bytes) tensor")
    _chk(scale_a, F32, "scale_a"); _chk(w_scale, F32, "w_scale")
    M, K = xq.shape
    N = wq.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=xq.device)
    if bias is not None:
        _chk(bias, BF16, "bias")
    _l.check(L.st355_linear_fp8(_stream(), _ptr(xq), K, _ptr(scale_a), _ptr(wq), K, _ptr(w_scale), _ptr(bias), _ptr(out), _rows(out, "out"),
                                M, N, K), "linear_fp8")
    return out


def _tn_operand(t, name: str):
    """a TN operand: 2-D [M, C] (row stride free) or a 3-D [B, rows, C] view of a joint buffer (row stride ld, segment stride a multiple of it):
    -> (pointer tensor, ld, logical rows M, C, seg_rows or 0, physical segment stride in rows or 0)"""
    _chk(t, BF16, name)
    if t.dim() == 2:
        return t, _rows(t, name), t.shape[0], t.shape[1], 0, 0
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) <= 0 or t.stride(0) % t.stride(1) != 0:
        raise _l.St355Error(f"gemm_tn: {name} must be [M, C] or a [B, rows, C] view whose batch stride is a whole number of rows")
    B, rows, Cn = t.shape
    return t, t.stride(1), B * rows, Cn, rows, t.stride(0) // t.stride(1)


def gemm_tn(Lm, R, out=None, accumulate: bool = False):
    """out[P,Q] (+)= Lm[M,P]^T @ R[M,Q]  — the weight-gradient form (dW = dY^T X).  M must be a multiple of 64 (zero-padded rows).  Either operand may be a
    3-D [B, rows, C] strided view (a stream's rows of a joint buffer, rows a multiple of 64): contracted in place (st355_gemm_tn_seg_bf16)."""
    L = _l.load()
    Lm, ldl, M, P, seg_a, str_a = _tn_operand(Lm, "L")
    R, ldr, M2, Q, seg_b, str_b = _tn_operand(R, "R")
    if M2 != M:
        raise _l.St355Error("gemm_tn: operands must share the contraction length")
    if out is None:
        if accumulate:
            raise _l.St355Error("gemm_tn: accumulate needs an output tensor")
        out = torch.empty(P, Q, dtype=BF16, device=Lm.device)
    _chk(out, BF16, "out")
    ws = _gemm_workspace(Lm.device)                     # fp32 split-K slabs: up to 7 slices of a 6144 x 1536 gradient (264 MB)
    if seg_a or seg_b:
        seg = seg_a or seg_b
        if seg_a and seg_b and seg_a != seg_b:
            raise _l.St355Error("gemm_tn: two segmented operands must share the segment length")
        _l.check(L.st355_gemm_tn_seg_bf16(_stream(), _ptr(Lm), ldl, str_a if seg_a else 0, _ptr(R), ldr, str_b if seg_b else 0, _ptr(out), _rows(out, "out"),
                                          M, seg, P, Q, 1 if accumulate else 0, _ptr(ws), ws.numel() * 4), "gemm_tn_seg_bf16")
        return out
    _l.check(L.st355_gemm_tn_bf16(_stream(), _ptr(Lm), ldl, _ptr(R), ldr, _ptr(out), _rows(out, "out"), M, P, Q,
                                  1 if accumulate else 0, _ptr(ws), ws.numel() * 4), "gemm_tn_bf16")
    return out


_colsum_ws = {}


def colsum_prod(a, out, b=None, rows_per_batch: Optional[int] = None, mode: int = 0, prev=None, shift=None, scale=None, accumulate: bool = False):
    """out[bi, :] (+)= sum over the rows of batch bi of a (* b).  out: fp32 2-D view [nb, N] (row stride free).  mode 1: see st355.h."""
    L = _l.load()
    _chk(a, BF16, "a"); _chk(out, F32, "out")
    rows, N = a.shape
    rpb = rows if rows_per_batch is None else rows_per_batch
    if b is not None:
        _chk(b, BF16, "b")
    need = L.st355_colsum_workspace(rows, N, rpb)
    ws = _colsum_ws.get(a.device.index)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=F32, device=a.device)
        _colsum_ws[a.device.index] = ws
    ms = 0
    if mode == 1:
        _chk(shift, BF16, "shift"); _chk(scale, BF16, "scale"); _chk(prev, F32, "prev")
        ms = _rows(shift, "shift")
        if _rows(scale, "scale") != ms:
            raise _l.St355Error("colsum_prod: shift and scale must share a row stride")
    _l.check(L.st355_colsum_prod(_stream(), _ptr(a), _rows(a, "a"), _ptr(b), _rows(b, "b") if b is not None else 0, rows, N, rpb, _ptr(out),
                                 _rows(out, "out"), mode, _ptr(prev), _rows(prev, "prev") if prev is not None else 0, _ptr(shift), _ptr(scale),
                                 ms, 1 if accumulate else 0, _ptr(ws)), "colsum_prod")
    return out


_stats_ws = {}


def _stats_workspace(dev, nbytes: int):
    ws = _stats_ws.get(dev.index)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _ws_retired.append(ws)
        ws = _stats_ws[dev.index] = torch.empty((nbytes + 3) // 4, dtype=F32, device=dev)
    return ws


def stat_out(out, reduce_batches: bool = False, accumulate: bool = False):
    """st355_stat_out of a destination tensor: fp32 2-D view [nb, N] (per-batch rows, row stride free) or — reduce_batches — one fp32 / bf16 row [N]"""
    o = _l.StatOut()
    if out is None:
        return o
    if out.dtype not in (F32, BF16) or (out.dtype == BF16 and not reduce_batches):
        raise _l.St355Error("stat_out: per-batch sums are fp32; a bf16 destination is one row summed over the batches")
    _dev(out, "stat_out")
    o.out, o.stride = out.data_ptr(), (0 if reduce_batches else _rows(out, "stat_out"))
    o.reduce_batches, o.out_bf16, o.accumulate = int(reduce_batches), int(out.dtype == BF16), int(accumulate)
    o._keep = out
    return o


def ln_modulate_bwd_stats(dy, x, scale, rows_per_batch: int, d_shift, d_scale, dres=None, gate=None, y_branch=None, d_gate=None, d_bias=None, eps: float = 1e-6,
                          want_gated: bool = False, out=None):
    """ln_modulate_bwd + the sums autograd accumulates around the AdaLN instance (st355_ln_modulate_bwd_stats): d_shift / d_scale fp32 [nb, D] views,
    d_gate = sum dx * y_branch (fp32 [nb, D]), d_bias = sum gate * dx over all rows (one fp32 / bf16 row).  Returns (dx, dxg)."""
    L = _l.load()
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(scale, BF16, "scale")
    rows, D = x.shape
    dx = torch.empty(rows, D, dtype=BF16, device=x.device) if out is None else out
    dxg = torch.empty(rows, D, dtype=BF16, device=x.device) if want_gated else None
    ws = _stats_workspace(x.device, L.st355_stats_workspace(rows, D, rows_per_batch, 4))
    outs = [stat_out(d_shift), stat_out(d_scale), stat_out(d_gate), stat_out(d_bias, reduce_batches=True)]
    _l.check(L.st355_ln_modulate_bwd_stats(_stream(), _ptr(dy), _rows(dy, "dy"), _ptr(x), _rows(x, "x"), _ptr(scale), _rows(scale, "scale"), rows_per_batch,
                                           _ptr(dres), _rows(dres, "dres") if dres is not None else 0, _ptr(gate) if want_gated else None,
                                           _rows(gate, "gate") if want_gated else 0, _ptr(dx), _rows(dx, "dx"), _ptr(dxg), D, rows, D, eps,
                                           _ptr(y_branch), _rows(y_branch, "y_branch") if y_branch is not None else 0,
                                           C.byref(outs[0]), C.byref(outs[1]), C.byref(outs[2]), C.byref(outs[3]), _ptr(ws)), "ln_modulate_bwd_stats")
    return dx, dxg


def scale_cols_stats(x, gate, rows_per_batch: int, y_branch=None, d_gate=None, d_bias=None, out=None):
    """scale_cols + d_gate = sum_t x * y_branch (fp32 [nb, N]) + d_bias = sum over all rows of the output (one fp32 / bf16 row)"""
    L = _l.load()
    _chk(x, BF16, "x"); _chk(gate, BF16, "gate")
    M, N = x.shape
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=x.device)
    ws = _stats_workspace(x.device, L.st355_stats_workspace(M, N, rows_per_batch, 2))
    og, ob = stat_out(d_gate), stat_out(d_bias, reduce_batches=True)
    _l.check(L.st355_scale_cols_stats(_stream(), _ptr(x), _rows(x, "x"), _ptr(gate), _rows(gate, "gate"), rows_per_batch, _ptr(out), _rows(out, "out"), M, N,
                                      _ptr(y_branch), _rows(y_branch, "y_branch") if y_branch is not None else 0, C.byref(og), C.byref(ob), _ptr(ws)),
             "scale_cols_stats")
    return out


def colsum_rows(a, rows_per_batch: int, batch_stride_rows: int, nb: int, out, per_batch: bool = False, accumulate: bool = False):
    """column sums of nb row blocks of `a` (block b = rows [b * batch_stride_rows, + rows_per_batch) of the 2-D tensor a, in place): one row over all blocks
    (fp32 / bf16 [N]) or per-block fp32 rows [nb, N]"""
    L = _l.load()
    _chk(a, BF16, "a")
    N = a.shape[1]
    ws = _stats_workspace(a.device, L.st355_stats_workspace(nb * rows_per_batch, N, rows_per_batch, 1))
    o = stat_out(out, reduce_batches=not per_batch, accumulate=accumulate)
    _l.check(L.st355_colsum_rows(_stream(), _ptr(a), _rows(a, "a"), rows_per_batch, batch_stride_rows, nb, N, C.byref(o), _ptr(ws)), "colsum_rows")
    return out


def transpose(src, out=None):
    """out[c, r] = src[r, c] (bf16)"""
    L = _l.load()
    _chk(src, BF16, "src")
    R, Cn = src.shape
    if out is None:
        out = torch.empty(Cn, R, dtype=BF16, device=src.device)
    _chk(out, BF16, "out")
    _l.check(L.st355_transpose_bf16(_stream(), _ptr(src), _rows(src, "src"), _ptr(out), _rows(out, "out"), R, Cn), "transpose_bf16")
    return out


_skinny_ws = {}


def skinny_tn(Lm, R, out, so_p: int, so_r: int, r_used: int, alpha: float = 1.0, accumulate: bool = False):
    """out[p*so_p + r*so_r] (+)= alpha * sum_m Lm[m,p] * R[m,r]   (fp32 out; rank-space LoRA gradients).  Lm / R may be 3-D segmented views (_seg)."""
    L = _l.load()
    _chk(Lm, BF16, "L"); _chk(R, BF16, "R"); _chk(out, F32, "out")
    M, P, ldl, seg, seg_l = _seg(Lm, "L")
    Mr, Rn, ldr, sr, seg_r = _seg(R, "R")
    if Mr != M:
        raise _l.St355Error(f"skinny_tn: L has {M} rows, R has {Mr}")
    seg = _seg_join(seg, sr, "skinny_tn")
    need = L.st355_skinny_tn_workspace(M, P, Rn)
    key = (Lm.device.index,)
    ws = _skinny_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=F32, device=Lm.device)
        _skinny_ws[key] = ws
    _l.check(L.st355_skinny_tn_seg(_stream(), _ptr(Lm), ldl, _ptr(R), ldr, _ptr(out), so_p, so_r, M, P, Rn,
                                   r_used, alpha, 1 if accumulate else 0, _ptr(ws), seg, seg_l, seg_r), "skinny_tn")
    return out


def skinny_tn_multi(Lm, R, outs, so_p: int, so_r: int, r_used: int, alpha: float = 1.0, accumulate: bool = False):
    """outs[g][p*so_p + r*so_r] (+)= alpha * sum_m Lm[m,p] * R[m, 32 g + r] for the len(outs) <= 4 adapters that share Lm: ONE pass over Lm (R: [M, >= 128])."""
    L = _l.load()
    _chk(Lm, BF16, "L"); _chk(R, BF16, "R")
    M, P, ldl, seg, seg_l = _seg(Lm, "L")
    Mr, Rn, ldr, sr, seg_r = _seg(R, "R")
    if Mr != M or Rn < 128 or not (1 <= len(outs) <= 4):
        raise _l.St355Error(f"skinny_tn_multi: L has {M} rows, R {Mr} x {Rn} (needs 128 columns), {len(outs)} outputs (1..4)")
    seg = _seg_join(seg, sr, "skinny_tn_multi")
    for o in outs:
        _chk(o, F32, "out")
    need = L.st355_skinny_tn_workspace(M, P, 128)
    key = (Lm.device.index,)
    ws = _skinny_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=F32, device=Lm.device)
        _skinny_ws[key] = ws
    arr = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
    _l.check(L.st355_skinny_tn_multi(_stream(), _ptr(Lm), ldl, _ptr(R), ldr, arr, len(outs), so_p, so_r, M, P, r_used, alpha, 1 if accumulate else 0,
                                     _ptr(ws), seg, seg_l, seg_r), "skinny_tn_multi")
    return outs


# ------------------------------------------------------------------------------------------------
# AdaLN / RMSNorm + RoPE
# ------------------------------------------------------------------------------------------------
def ln_modulate_fwd(x, scale, shift, rows_per_batch: int, eps: float = 1e-6, out=None):
    """y = LN(x) * (1 + scale[b]) + shift[b]; scale/shift are [B, D] views sharing one row stride."""
    L = _l.load()
    _chk(x, BF16, "x"); _chk(scale, BF16, "scale"); _chk(shift, BF16, "shift")
    rows, D = x.shape
    ms = _rows(scale, "scale")
    if _rows(shift, "shift") != ms:
        raise _l.St355Error("ln_modulate_fwd: scale and shift must share a row stride")
    if out is None:
        out = torch.empty(rows, D, dtype=BF16, device=x.device)
    _l.check(L.st355_ln_modulate_fwd(_stream(), _ptr(x), _rows(x, "x"), _ptr(scale), _ptr(shift), ms, rows_per_batch,
                                     _ptr(out), _rows(out, "out"), rows, D, eps), "ln_modulate_fwd")
    return out


_zero_rows = {}


def layer_norm_xhat(x, eps: float = 1e-6, out=None):
    """xhat = LN(x) with no modulation (ln_modulate_fwd with scale = shift = 0): the factor of the modulation-SCALE gradient, d scale_b = sum_t dY * xhat.
    The engines compute that sum from xhat itself; recovering xhat from the saved modulated output as (n - shift) / (1 + scale) (colsum_prod mode 1) is
    singular where a scale entry is exactly -1 in bf16 — which random-init modulation tables hit in a few entries per step."""
    _chk(x, BF16, "x")
    key = (str(x.device), x.shape[1])
    z = _zero_rows.get(key)
    if z is None:
        z = _zero_rows[key] = torch.zeros(1, x.shape[1], dtype=BF16, device=x.device)
    return ln_modulate_fwd(x, z, z, x.shape[0], eps=eps, out=out)


def ln_modulate_bwd(dy, x, scale, rows_per_batch: int, dres=None, gate=None, eps: float = 1e-6, want_gated: bool = False, out=None):
    """dx = dres + LNbwd(dy*(1+scale));  dxg = gate[b]*dx (if want_gated).  Returns (dx, dxg).  out: optional destination view for dx (row stride free)."""
    L = _l.load()
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(scale, BF16, "scale")
    rows, D = x.shape
    dx = torch.empty(rows, D, dtype=BF16, device=x.device) if out is None else out
    if out is not None:
        _chk(out, BF16, "out")
        if tuple(out.shape) != (rows, D):
            raise _l.St355Error("ln_modulate_bwd: out shape mismatch")
    dxg = torch.empty(rows, D, dtype=BF16, device=x.device) if want_gated else None
    if dres is not None:
        _chk(dres, BF16, "dres")
    if want_gated:
        _chk(gate, BF16, "gate")
    _l.check(L.st355_ln_modulate_bwd(_stream(), _ptr(dy), _rows(dy, "dy"), _ptr(x), _rows(x, "x"), _ptr(scale), _rows(scale, "scale"),
                                     rows_per_batch, _ptr(dres), _rows(dres, "dres") if dres is not None else 0,
                                     _ptr(gate) if want_gated else None, _rows(gate, "gate") if want_gated else 0,
                                     _ptr(dx), _rows(dx, "dx"), _ptr(dxg), D, rows, D, eps), "ln_modulate_bwd")
    return dx, dxg


def qk_norm_rope_fwd(qkv, wq, wk, cos, sin, Q, K, Qt, Kt, Vt, B, H, d, S_part, pos0, S, Sp, eps: float = 1e-6):
    L = _l.load()
    _chk(qkv, BF16, "qkv"); _chk(cos, F32, "cos"); _chk(sin, F32, "sin")
    _l.check(L.st355_qk_norm_rope_fwd(_stream(), _ptr(qkv), _rows(qkv, "qkv"), _ptr(wq), _ptr(wk), _ptr(cos), _ptr(sin),
                                      _ptr(Q), _ptr(K), _ptr(Qt), _ptr(Kt), _ptr(Vt), B, H, d, S_part, pos0, S, Sp, eps),
             "qk_norm_rope_fwd")


def qk_norm_rope_bwd(dQ, dK, qkv, wq, wk, cos, sin, dqkv, B, H, d, S_part, pos0, S, eps: float = 1e-6):
    L = _l.load()
    _chk(dQ, BF16, "dQ"); _chk(dK, BF16, "dK"); _chk(qkv, BF16, "qkv"); _chk(dqkv, BF16, "dqkv")
    _l.check(L.st355_qk_norm_rope_bwd(_stream(), _ptr(dQ), _ptr(dK), _ptr(qkv), _rows(qkv, "qkv"), _ptr(wq), _ptr(wk),
                                      _ptr(cos), _ptr(sin), _ptr(dqkv), _rows(dqkv, "dqkv"), B, H, d, S_part, pos0, S, eps),
             "qk_norm_rope_bwd")


_qkwg_ws = {}


def qk_norm_rope_bwd_wgrad(dQ, dK, qkv, wq, wk, cos, sin, dqkv, B, H, d, S_part, pos0, S, gwq, gwk, accumulate: bool = False, eps: float = 1e-6):
    """qk_norm_rope_bwd + d loss / d (norm_q.weight, norm_k.weight) into the bf16 [d] views gwq / gwk (None: absent / frozen)"""
    L = _l.load()
    _chk(dQ, BF16, "dQ"); _chk(dK, BF16, "dK"); _chk(qkv, BF16, "qkv"); _chk(dqkv, BF16, "dqkv")
    for g in (gwq, gwk):
        if g is not None:
            _chk(g, BF16, "gw")
    need = L.st355_qk_norm_wgrad_workspace(B, H, d, S_part)
    ws = _qkwg_ws.get(dQ.device.index)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=F32, device=dQ.device)
        _qkwg_ws[dQ.device.index] = ws
    _l.check(L.st355_qk_norm_rope_bwd_wgrad(_stream(), _ptr(dQ), _ptr(dK), _ptr(qkv), _rows(qkv, "qkv"), _ptr(wq), _ptr(wk), _ptr(cos), _ptr(sin),
                                            _ptr(dqkv), _rows(dqkv, "dqkv"), B, H, d, S_part, pos0, S, eps, _ptr(gwq), _ptr(gwk),
                                            1 if accumulate else 0, _ptr(ws)), "qk_norm_rope_bwd_wgrad")


def qk_rope_norm_bwd(dQ, dK, Q, K, rrms, wq, wk, cos, sin, dqkv, B, H, d, S_part, pos0, S):
    """backward of the fused projection epilogue: from the roped head-major Q / K and the saved 1/rms (no pre-norm projection kept)"""
    L = _l.load()
    _chk(dQ, BF16, "dQ"); _chk(dK, BF16, "dK"); _chk(Q, BF16, "Q"); _chk(K, BF16, "K"); _chk(rrms, F32, "rrms"); _chk(dqkv, BF16, "dqkv")
    _l.check(L.st355_qk_rope_norm_bwd(_stream(), _ptr(dQ), _ptr(dK), _ptr(Q), _ptr(K), _ptr(rrms), _ptr(wq), _ptr(wk), _ptr(cos), _ptr(sin),
                                      _ptr(dqkv), _rows(dqkv, "dqkv"), B, H, d, S_part, pos0, S), "qk_rope_norm_bwd")


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _res_ok(O, O_res):
    _chk(O_res, BF16, "O_res")
    if O_res.shape != O.shape or _rows(O_res, "O_res") != _rows(O, "O"):
        raise _l.St355Error("attention: O_res must have O's shape and row stride")


def attn_fwd(Q, K, Vt, O, lse2, B, H, S, Sp, d, scale: float, key_bias=None, O_res=None):
    """O: 2-D token-major view [B*S, >=H*d]; lse2 [B,H,S] fp32.  O_res (same layout): also write the rounding residual bf16(O_fp32 - O) for attn_bwd(O_res=...)."""
    L = _l.load()
    _chk(Q, BF16, "Q"); _chk(K, BF16, "K"); _chk(Vt, BF16, "Vt"); _chk(O, BF16, "O"); _chk(lse2, F32, "lse2")
    if O_res is not None:
        _res_ok(O, O_res)
        _l.check(L.st355_attn_fwd_res(_stream(), _ptr(Q), _ptr(K), _ptr(Vt), _ptr(key_bias), _ptr(O), _rows(O, "O"), _ptr(O_res), _ptr(lse2),
                                      B, H, S, S, Sp, d, scale), "attn_fwd_res")
        return
    _l.check(L.st355_attn_fwd(_stream(), _ptr(Q), _ptr(K), _ptr(Vt), _ptr(key_bias), _ptr(O), _rows(O, "O"), _ptr(lse2),
                              B, H, S, Sp, d, scale), "attn_fwd")


def attn_fwd_vrows(Q, K, v_rows, O, lse2, B, H, S, d, scale: float, key_bias=None):
    """attn_fwd with V row-major: v_rows is a 2-D token-major view [B*S, >= H*d] (head h at columns h*d); no V^T copy.  d = 128."""
    L = _l.load()
    _chk(Q, BF16, "Q"); _chk(K, BF16, "K"); _chk(v_rows, BF16, "v_rows"); _chk(O, BF16, "O"); _chk(lse2, F32, "lse2")
    _l.check(L.st355_attn_fwd_vrows(_stream(), _ptr(Q), _ptr(K), _ptr(v_rows), _rows(v_rows, "v_rows"), _ptr(key_bias), _ptr(O), _rows(O, "O"),
                                    _ptr(lse2), B, H, S, d, scale), "attn_fwd_vrows")


_attn_ws = {}


# ST355_ATTN_TR=0: the engines build head-major Q^T / K^T copies and the backward reads them (dkv2 / dq); default: no copies, transposing LDS reads
ATTN_TR = os.environ.get("ST355_ATTN_TR", "1") != "0"


def attn_bwd(Q, K, Qt, Kt, v_rows, O, dO, lse2, dQ, dK, dv_rows, B, H, S, Sp, d, scale: float, key_bias=None, O_res=None):
    L = _l.load()
    need = L.st355_attn_bwd_workspace(B, H, S, Sp, d)
    key = (Q.device.index,)
    ws = _attn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=Q.device)
        _attn_ws[key] = ws
    if O_res is not None:          # delta = rowsum(dO * (O + O_res))
        _res_ok(O, O_res)
        _l.check(L.st355_attn_bwd_res(_stream(), _ptr(Q), _ptr(K), _ptr(Qt), _ptr(Kt), _ptr(v_rows), _rows(v_rows, "v_rows"),
                                      _ptr(O), _rows(O, "O"), _ptr(O_res), _ptr(dO), _rows(dO, "dO"), _ptr(lse2), _ptr(key_bias),
                                      _ptr(dQ), _ptr(dK), _ptr(dv_rows), _rows(dv_rows, "dv_rows"), B, H, S, Sp, S, Sp, d, scale, _ptr(ws)), "attn_bwd_res")
        return
    _l.check(L.st355_attn_bwd(_stream(), _ptr(Q), _ptr(K), _ptr(Qt), _ptr(Kt), _ptr(v_rows), _rows(v_rows, "v_rows"),
                              _ptr(O), _rows(O, "O"), _ptr(dO), _rows(dO, "dO"), _ptr(lse2), _ptr(key_bias),
                              _ptr(dQ), _ptr(dK), _ptr(dv_rows), _rows(dv_rows, "dv_rows"), B, H, S, Sp, d, scale, _ptr(ws)),
             "attn_bwd")


def attn_bwd_rope(Q, K, v_rows, O, dO, lse2, rrms, wq_lo, wk_lo, wq_hi, wk_hi, split: int, cos_p, sin_p, dqkv, B, H, S, Sp, d, scale: float, key_bias=None):
    """attention backward with the RoPE + RMSNorm backward fused into the dQ / dK epilogues (head_dim 128, after a fused QKV projection): dq, dk, dv are
    written straight into the rows of the projection gradient dqkv [B*S, >= 3*H*d]; joint positions < split use the *_lo norm weights."""
    L = _l.load()
    _chk(Q, BF16, "Q"); _chk(K, BF16, "K"); _chk(rrms, F32, "rrms"); _chk(dqkv, BF16, "dqkv"); _chk(cos_p, F32, "cos_p"); _chk(sin_p, F32, "sin_p")
    need = L.st355_attn_bwd_workspace(B, H, S, Sp, d)
    key = (Q.device.index,)
    ws = _attn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=Q.device)
        _attn_ws[key] = ws
    _l.check(L.st355_attn_bwd_rope(_stream(), _ptr(Q), _ptr(K), _ptr(v_rows), _rows(v_rows, "v_rows"), _ptr(O), _rows(O, "O"), _ptr(dO), _rows(dO, "dO"),
                                   _ptr(lse2), _ptr(key_bias), _ptr(rrms), _ptr(wq_lo), _ptr(wk_lo), _ptr(wq_hi), _ptr(wk_hi), split, _ptr(cos_p), _ptr(sin_p),
                                   _ptr(dqkv), _rows(dqkv, "dqkv"), B, H, S, Sp, d, scale, _ptr(ws)), "attn_bwd_rope")


# ------------------------------------------------------------------------------------------------
# optimiser / EMA
# ------------------------------------------------------------------------------------------------
def adamw_ema_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2,
                   grad_scale: float = 1.0, ema=None, ema_decay: float = 0.0, p_bf16=None):
    L = _l.load()
    if p.dtype == F32:
        for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
            _chk(t, F32, n)
        _l.check(L.st355_adamw_ema_step(_stream(), _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema), _ptr(p_bf16), p.numel(),
                                        lr, beta1, beta2, eps, weight_decay, step, grad_scale, ema_decay), "adamw_ema_step")
    else:
        _chk(p, BF16, "p"); _chk(g, BF16, "g"); _chk(m, F32, "m"); _chk(v, F32, "v")
        _l.check(L.st355_adamw_ema_step_bf16(_stream(), _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(ema), p.numel(),
                                             lr, beta1, beta2, eps, weight_decay, step, grad_scale, ema_decay),
                 "adamw_ema_step_bf16")


def ema_update(shadow, param, decay: float):
    L = _l.load()
    if shadow.dtype != param.dtype:
        raise _l.St355Error("ema_update: dtype mismatch")
    _l.check(L.st355_ema_update(_stream(), _ptr(shadow), _ptr(param), shadow.numel(), decay, shadow.element_size()), "ema_update")


_gn_ws = {}


def grad_norm(g):
    """returns fp32 [2] device tensor: (sum of squares, max |g|).  The per-block partials live in a scratch of this (device, stream): norms on different streams
    (a side-stream EMA / ControlNet norm, a hipGraph capture next to eager calls) never share one"""
    L = _l.load()
    out = torch.empty(2, dtype=F32, device=g.device)
    st = _stream()
    key = (g.device.index, int(st) if st is not None else 0)
    ws = _gn_ws.get(key)
    if ws is None:
        ws = _gn_ws[key] = torch.empty(2 * 1024, dtype=F32, device=g.device)
    _l.check(L.st355_grad_norm_ws(st, _ptr(g), g.numel(), g.element_size(), _ptr(out), _ptr(ws)), "grad_norm")
    return out


def grad_clamp_(g, c: float):
    """in place: g <- clamp(g, -c, c) (clip_grad_value_)"""
    L = _l.load()
    _dev(g, "g")
    if g.dtype not in (F32, BF16) or not g.is_contiguous():
        raise _l.St355Error("grad_clamp: expected a contiguous fp32 / bf16 tensor")
    _l.check(L.st355_grad_clamp(_stream(), _ptr(g), g.numel(), g.element_size(), float(c)), "grad_clamp")
    return g


def grad_clip_norm_(g, stats, max_norm: float, pre_scale: float = 1.0):
    """in place: g *= min(1, max_norm / (sqrt(stats[0]) * pre_scale + 1e-6)); stats = grad_norm(g) (device, never read on the host)"""
    L = _l.load()
    _dev(g, "g"); _chk(stats, F32, "stats")
    if g.dtype not in (F32, BF16) or not g.is_contiguous():
        raise _l.St355Error("grad_clip_norm: expected a contiguous fp32 / bf16 tensor")
    _l.check(L.st355_grad_clip_norm(_stream(), _ptr(g), g.numel(), g.element_size(), _ptr(stats), float(max_norm), float(pre_scale)), "grad_clip_norm")
    return g


def lora_pack(A, Bm, scale: float, A_cat, A_cat_T, B_blk, B_blk_T, k2_off: int = 0, n_off: int = 0):
    """write one adapter (A [r,K], B [N,r], fp32) into the block-structured bf16 operands of a fused projection group"""
    L = _l.load()
    _chk(A, F32, "A"); _chk(Bm, F32, "B")
    r, K = A.shape
    N = Bm.shape[0]
    K2 = A_cat.shape[0]
    N_total = B_blk.shape[0]
    if A_cat.shape != (K2, K) or A_cat_T.shape != (K, K2) or B_blk.shape != (N_total, K2) or B_blk_T.shape != (K2, N_total):
        raise _l.St355Error("lora_pack: operand shapes inconsistent")
    for t in (A_cat, A_cat_T, B_blk, B_blk_T):
        if not t.is_contiguous():
            raise _l.St355Error("lora_pack: packed operands must be contiguous")
    _l.check(L.st355_lora_pack(_stream(), _ptr(A.contiguous()), _ptr(Bm.contiguous()), r, K, N, scale, _ptr(A_cat), _ptr(A_cat_T),
                               _ptr(B_blk), _ptr(B_blk_T), K2, k2_off, N_total, n_off), "lora_pack")


def adamw_bf16_sr_step(p, g, m, v, shift, step: int, lr: float, beta1: float, beta2: float, eps: float, seg_end=None, seg_decay=None,
                       rand_bits=None, seed: int = 0, offset: int = 0, grad_scale: float = 1.0):
    """AdamWBF16.step over flat bf16 arenas (p, exp_avg, exp_avg_sq, shift updated in place).  seg_end int64 / seg_decay fp32 device
    arrays describe the parameter tensors inside the arena and the decay released for each this step; rand_bits int32 [4, n] injects the
    stochastic-rounding draws (tests), else in-kernel Philox."""
    L = _l.load()
    for t, nm in ((p, "p"), (g, "g"), (m, "exp_avg"), (v, "exp_avg_sq"), (shift, "shift")):
        _chk(t, BF16, nm)
        if not t.is_contiguous() or t.numel() != p.numel():
            raise _l.St355Error(f"adamw_bf16_sr_step: {nm} must be a contiguous arena of the same length")
    nseg = 0
    if seg_end is not None:
        if seg_end.dtype != torch.int64 or seg_decay.dtype != F32 or seg_end.numel() != seg_decay.numel():
            raise _l.St355Error("adamw_bf16_sr_step: seg_end must be int64 and seg_decay fp32, same length")
        _dev(seg_end, "seg_end"); _dev(seg_decay, "seg_decay")
        nseg = seg_end.numel()
    if rand_bits is not None:
        if rand_bits.dtype != torch.int32 or rand_bits.numel() != 4 * p.numel() or not rand_bits.is_contiguous():
            raise _l.St355Error("adamw_bf16_sr_step: rand_bits must be a contiguous int32 [4, n] tensor")
        _dev(rand_bits, "rand_bits")
    _l.check(L.st355_adamw_bf16_sr_step(_stream(), _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(shift), p.numel(), int(step), float(lr),
                                        float(beta1), float(beta2), float(eps), _ptr(seg_end), _ptr(seg_decay), nseg, _ptr(rand_bits),
                                        int(seed), int(offset), float(grad_scale)), "adamw_bf16_sr_step")


# ------------------------------------------------------------------------------------------------
# profiler
# ------------------------------------------------------------------------------------------------
def prof_enable(on: bool):
    _l.load().st355_prof_enable(1 if on else 0)


def prof_reset():
    _l.load().st355_prof_reset()


def prof_dump(path: str):
    """one CSV line per recorded launch (class index, ms, algorithmic flops / bytes, shape tag)"""
    _l.check(_l.load().st355_prof_dump(str(path).encode()), "prof_dump")


def prof_collect():
    n = len(_l.KERNEL_CLASSES)
    ms = (C.c_double * n)(); la = (C.c_int64 * n)(); fl = (C.c_double * n)(); by = (C.c_double * n)()
    _l.load().st355_prof_collect(ms, la, fl, by, n)
    return {k: {"ms": ms[i], "launches": la[i], "flops": fl[i], "bytes": by[i]} for i, k in enumerate(_l.KERNEL_CLASSES)}


# ------------------------------------------------------------------------------------------------
# UNet path: grid buffers, convolution-as-GEMM, GroupNorm, GEGLU, affine LayerNorm, cross-attention
# ------------------------------------------------------------------------------------------------
def grid_rows(B: int, H: int, W: int) -> int:
    return int(_l.load().st355_conv_grid_rows(B, H, W))


_grid_pool = {}


def grid_zeros(B: int, H: int, W: int, C_: int, device, pool: bool = False):
    """a zero-filled grid buffer [grid_rows, C] (border + 64 tail rows stay zero: kernels never write them non-zero)"""
    return torch.zeros(grid_rows(B, H, W), C_, dtype=BF16, device=device)


def _grid_out(B: int, H: int, W: int, C_: int, device, conv: bool = False):
    """output buffer of a grid kernel WITHOUT the full zero-fill pass: only the rows the kernel does not write are zeroed —
    the 64 tail rows for the layout kernels (they write every grid position, borders as zero), plus the first / last W+3 border positions for
    st355_conv_bf16 (its GEMM covers rows [W+3, rows - W - 3))."""
    n = B * (H + 2) * (W + 2)
    t = torch.empty(n + 64, C_, dtype=BF16, device=device)
    if conv:
        t[:W + 3].zero_()
        t[n - (W + 3):].zero_()
    else:
        t[n:].zero_()
    return t


def grid_from_nchw(x, Cpad: int):
    L = _l.load()
    _chk(x, BF16, "x")
    B, Cn, H, W = x.shape
    g = _grid_out(B, H, W, Cpad, x.device)
    _l.check(L.st355_grid_from_nchw(_stream(), _ptr(x.contiguous()), _ptr(g), B, Cn, H, W, Cpad), "grid_from_nchw")
    return g


def grid_to_nchw(g, B: int, Cn: int, H: int, W: int):
    L = _l.load()
    _chk(g, BF16, "grid")
    y = torch.empty(B, Cn, H, W, dtype=BF16, device=g.device)
    _l.check(L.st355_grid_to_nchw(_stream(), _ptr(g), _ptr(y), B, Cn, H, W, g.shape[1]), "grid_to_nchw")
    return y


def conv(x, w, B: int, H: int, W: int, bias=None, img_add=None, residual=None, taps: int = 9, out=None):
    """grid conv (3x3 stride 1 pad 1 for taps=9; 1x1 / pre-gathered columns for taps=1).  x: grid [.., Cin]; w: [Cout, taps*Cin]"""
    L = _l.load()
    _chk(x, BF16, "x"); _chk(w, BF16, "w")
    Cin = x.shape[1]
    Cout = w.shape[0]
    if w.shape[1] != taps * Cin or not w.is_contiguous() or not x.is_contiguous():
        raise _l.St355Error(f"conv: weight {tuple(w.shape)} does not match taps*Cin = {taps}*{Cin} (or non-contiguous operands)")
    if x.shape[0] != grid_rows(B, H, W):
        raise _l.St355Error(f"conv: x has {x.shape[0]} rows, a ({B},{H},{W}) grid has {grid_rows(B, H, W)}")
    if out is None:
        out = _grid_out(B, H, W, Cout, x.device, conv=True)
    if bias is not None:
        _chk(bias, BF16, "bias")
    if img_add is not None:
        _chk(img_add, BF16, "img_add")
    if residual is not None:
        _chk(residual, BF16, "residual")
    _l.check(L.st355_conv_bf16(_stream(), _ptr(x), _ptr(w), _ptr(bias), _ptr(img_add), _rows(img_add, "img_add") if img_add is not None else 0,
                               _ptr(residual), _ptr(out), B, H, W, Cin, Cout, taps), "conv_bf16")
    return out


def conv_wgrad(x, dy, dw, B: int, H: int, W: int, taps: int = 9, accumulate: bool = False):
    L = _l.load()
    _chk(x, BF16, "x"); _chk(dy, BF16, "dy"); _chk(dw, BF16, "dw")
    Cin, Cout = x.shape[1], dy.shape[1]
    if tuple(dw.shape) != (Cout, taps * Cin) or not dw.is_contiguous():
        raise _l.St355Error(f"conv_wgrad: dw must be a contiguous [{Cout}, {taps * Cin}] tensor")
    ws = _gemm_workspace(x.device)
    _l.check(L.st355_conv_wgrad_bf16(_stream(), _ptr(x), _ptr(dy), _ptr(dw), B, H, W, Cin, Cout, taps, 1 if accumulate else 0, _ptr(ws),
                                     ws.numel() * 4), "conv_wgrad_bf16")
    return dw


def im2col3x3(x, B: int, H: int, W: int, stride: int = 1, pad: int = 1):
    L = _l.load()
    _chk(x, BF16, "x")
    Cn = x.shape[1]
    Kpad = (9 * Cn + 63) // 64 * 64
    col = _grid_out(B, H // stride, W // stride, Kpad, x.device)
    _l.check(L.st355_im2col3x3(_stream(), _ptr(x), _ptr(col), B, H, W, Cn, stride, Kpad, pad), "im2col3x3")
    return col


def col2im3x3(dcol, B: int, H: int, W: int, Cn: int, stride: int = 1, pad: int = 1):
    L = _l.load()
    _chk(dcol, BF16, "dcol")
    dx = _grid_out(B, H, W, Cn, dcol.device)
    _l.check(L.st355_col2im3x3(_stream(), _ptr(dcol), _ptr(dx), B, H, W, Cn, stride, dcol.shape[1], pad), "col2im3x3")
    return dx


def upsample2x(x, B: int, H: int, W: int):
    L = _l.load()
    _chk(x, BF16, "x")
    y = _grid_out(B, 2 * H, 2 * W, x.shape[1], x.device)
    _l.check(L.st355_upsample2x(_stream(), _ptr(x), _ptr(y), B, H, W, x.shape[1]), "upsample2x")
    return y


def upsample2x_bwd(dy, B: int, H: int, W: int):
    L = _l.load()
    _chk(dy, BF16, "dy")
    dx = _grid_out(B, H, W, dy.shape[1], dy.device)
    _l.check(L.st355_upsample2x_bwd(_stream(), _ptr(dy), _ptr(dx), B, H, W, dy.shape[1]), "upsample2x_bwd")
    return dx


BLOCK_CALLS = {}          # block-level entry point -> how often it ran (the GPU suite asserts that the engines really take them)


def _fill(st, **kw):
    """tensors -> device pointers, None -> NULL, numbers as they are"""
    keep = []
    names = {f[0] for f in type(st)._fields_}
    for k, v in kw.items():
        if k not in names:
            raise _l.St355Error(f"{type(st).__name__}: no field {k!r} (include/st355.h)")
        if torch.is_tensor(v):
            _dev(v, k)
            keep.append(v)
            v = v.data_ptr()
        setattr(st, k, v)
    st._keep = keep
    return st


def block_flux_single_fwd(**kw):
    """st355_block_flux_single_fwd: one FluxSingleTransformerBlock forward as ONE C call (field names of st355_flux_single_fwd_args)"""
    L = _l.load()
    ws = _gemm_workspace(kw["x"].device)
    a = _fill(_l.FluxSingleFwdArgs(), gemm_ws=ws, gemm_ws_bytes=ws.numel() * 4, **kw)
    _l.check(L.st355_block_flux_single_fwd(_stream(), C.byref(a)), "block_flux_single_fwd")
    BLOCK_CALLS["block_flux_single_fwd"] = BLOCK_CALLS.get("block_flux_single_fwd", 0) + 1


def block_flux_double_fwd(**kw):
    """st355_block_flux_double_fwd: one FluxTransformerBlock forward as ONE C call (field names of st355_flux_double_fwd_args)"""
    L = _l.load()
    ws = _gemm_workspace(kw["img"].device)
    a = _fill(_l.FluxDoubleFwdArgs(), gemm_ws=ws, gemm_ws_bytes=ws.numel() * 4, **kw)
    _l.check(L.st355_block_flux_double_fwd(_stream(), C.byref(a)), "block_flux_double_fwd")
    BLOCK_CALLS["block_flux_double_fwd"] = BLOCK_CALLS.get("block_flux_double_fwd", 0) + 1


def block_flux_double_bwd(grads, **kw):
    """st355_block_flux_double_bwd (field names of st355_flux_double_bwd_args); grads: {"gA_qkv": [...], "gB_qkv": [...], "gA_out": [...], "gB_out": [...]}"""
    L = _l.load()
    dev = kw["img"].device
    B, S, H, D = kw["B"], kw["Si"] + kw["St"], kw["H"], kw["D"]
    ws = _gemm_workspace(dev)
    need = L.st355_attn_bwd_workspace(B, H, S, S, 128)
    aws = _attn_ws.get((dev.index,))
    if aws is None or aws.numel() < need:
        aws = _attn_ws[(dev.index,)] = torch.empty(need, dtype=torch.uint8, device=dev)
    sws = None
    if kw.get("K2_qkv") or kw.get("K2_out"):
        need = L.st355_skinny_tn_workspace(B * kw["Si"], D, 128)
        sws = _skinny_ws.get((dev.index,))
        if sws is None or sws.numel() * 4 < need:
            sws = _skinny_ws[(dev.index,)] = torch.empty((need + 3) // 4, dtype=F32, device=dev)
    a = _fill(_l.FluxDoubleBwdArgs(), gemm_ws=ws, gemm_ws_bytes=ws.numel() * 4, attn_ws=aws, skinny_ws=sws, **kw)
    for name, ts in (grads or {}).items():
        arr = getattr(a, name)
        for i, t in enumerate(ts or []):
            _chk(t, F32, name); arr[i] = t.data_ptr()
    _l.check(L.st355_block_flux_double_bwd(_stream(), C.byref(a)), "block_flux_double_bwd")
    BLOCK_CALLS["block_flux_double_bwd"] = BLOCK_CALLS.get("block_flux_double_bwd", 0) + 1


def block_flux_single_bwd(gA, gB, **kw):
    """st355_block_flux_single_bwd (field names of st355_flux_single_bwd_args); gA / gB: lists of the adapters' fp32 gradient views"""
    L = _l.load()
    dev = kw["x"].device
    B, S, H, D = kw["B"], kw["S"], kw["H"], kw["D"]
    ws = _gemm_workspace(dev)
    need = L.st355_attn_bwd_workspace(B, H, S, S, 128)
    aws = _attn_ws.get((dev.index,))
    if aws is None or aws.numel() < need:
        aws = _attn_ws[(dev.index,)] = torch.empty(need, dtype=torch.uint8, device=dev)
    sws = None
    if kw.get("K2"):
        need = L.st355_skinny_tn_workspace(B * S, D, 128)
        sws = _skinny_ws.get((dev.index,))
        if sws is None or sws.numel() * 4 < need:
            sws = _skinny_ws[(dev.index,)] = torch.empty((need + 3) // 4, dtype=F32, device=dev)
    a = _fill(_l.FluxSingleBwdArgs(), gemm_ws=ws, gemm_ws_bytes=ws.numel() * 4, attn_ws=aws, skinny_ws=sws, **kw)
    for i, t in enumerate(gA or []):
        _chk(t, F32, "gA"); a.gA[i] = t.data_ptr()
    for i, t in enumerate(gB or []):
        _chk(t, F32, "gB"); a.gB[i] = t.data_ptr()
    _l.check(L.st355_block_flux_single_bwd(_stream(), C.byref(a)), "block_flux_single_bwd")
    BLOCK_CALLS["block_flux_single_bwd"] = BLOCK_CALLS.get("block_flux_single_bwd", 0) + 1


def block_pixart_fwd(**kw):
    """st355_block_pixart_fwd: one PixArt BasicTransformerBlock(ada_norm_single) forward as ONE C call (field names of st355_pixart_block_fwd_args)"""
    L = _l.load()
    a = _fill(_l.PixartBlockFwdArgs(), **kw)
    _l.check(L.st355_block_pixart_fwd(_stream(), C.byref(a)), "block_pixart_fwd")
    BLOCK_CALLS["block_pixart_fwd"] = BLOCK_CALLS.get("block_pixart_fwd", 0) + 1


def block_pixart_bwd(**kw):
    """st355_block_pixart_bwd: the data path of that block's backward as ONE C call (field names of st355_pixart_block_bwd_args)"""
    L = _l.load()
    dev = kw["h"].device
    Sp = (kw["S"] + 63) // 64 * 64
    need = L.st355_attn_bwd_workspace(kw["B"], kw["H"], kw["S"], Sp, kw["d_pad"])
    aws = _attn_ws.get((dev.index,))
    if aws is None or aws.numel() < need:
        aws = _attn_ws[(dev.index,)] = torch.empty(need, dtype=torch.uint8, device=dev)
    a = _fill(_l.PixartBlockBwdArgs(), attn_ws=aws, **kw)
    _l.check(L.st355_block_pixart_bwd(_stream(), C.byref(a)), "block_pixart_bwd")
    BLOCK_CALLS["block_pixart_bwd"] = BLOCK_CALLS.get("block_pixart_bwd", 0) + 1


def block_sd3_joint_fwd(**kw):
    """st355_block_sd3_joint_fwd: one SD3 JointTransformerBlock forward as ONE C call (field names of st355_sd3_joint_fwd_args)"""
    L = _l.load()
    ws = _gemm_workspace(kw["img"].device)
    a = _fill(_l.Sd3JointFwdArgs(), gemm_ws=ws, gemm_ws_bytes=ws.numel() * 4, **kw)
    _l.check(L.st355_block_sd3_joint_fwd(_stream(), C.byref(a)), "block_sd3_joint_fwd")
    BLOCK_CALLS["block_sd3_joint_fwd"] = BLOCK_CALLS.get("block_sd3_joint_fwd", 0) + 1


def block_sd3_joint_bwd(**kw):
    """st355_block_sd3_joint_bwd: the data path of that block's backward as ONE C call (field names of st355_sd3_joint_bwd_args)"""
    L = _l.load()
    dev = kw["img"].device
    S = kw["Si"] + kw["St"]
    need = L.st355_attn_bwd_workspace(kw["B"], kw["H"], S, (S + 63) // 64 * 64, kw["hd"])
    aws = _attn_ws.get((dev.index,))
    if aws is None or aws.numel() < need:
        aws = _attn_ws[(dev.index,)] = torch.empty(need, dtype=torch.uint8, device=dev)
    ws = _gemm_workspace(dev)
    if kw.get("dmod_img") is not None:          # the fused-statistics form (full fine-tune): fp32 partial rows of its column sums
        rmax = max(kw["Si"], kw["St"])
        kw["stats_ws"] = _stats_workspace(dev, L.st355_stats_workspace(kw["B"] * rmax, 4 * kw["D"], rmax, 1))
    a = _fill(_l.Sd3JointBwdArgs(), gemm_ws=ws, gemm_ws_bytes=ws.numel() * 4, attn_ws=aws, **kw)
    _l.check(L.st355_block_sd3_joint_bwd(_stream(), C.byref(a)), "block_sd3_joint_bwd")
    BLOCK_CALLS["block_sd3_joint_bwd"] = BLOCK_CALLS.get("block_sd3_joint_bwd", 0) + 1


class VaeEncoderTable:
    """the architecture + device-pointer table st355_vae_encode walks (include/st355.h lists the order); keeps the tensors alive"""

    def __init__(self, in_channels: int, latent_channels: int, block_out_channels, layers_per_block: int, norm_num_groups: int, tensors):
        self.tensors = list(tensors)
        for t in self.tensors:
            _chk(t, BF16, "vae tensor")
            if not t.is_contiguous():
                raise _l.St355Error("vae_encode: table tensors must be contiguous")
        n = len(self.tensors)
        self._ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in self.tensors])
        st = _l.VaeEncoder()
        st.in_channels, st.latent_channels, st.n_levels, st.layers_per_block, st.norm_num_groups = in_channels, latent_channels, len(block_out_channels), layers_per_block, norm_num_groups
        for i, c in enumerate(block_out_channels):
            st.block_out_channels[i] = int(c)
        st.n_tensors, st.tensors = n, C.cast(self._ptrs, C.POINTER(C.c_void_p))
        self.struct = st


_vae_ws = {}


def vae_encode(table: VaeEncoderTable, x):
    """AutoencoderKL.encode as one C call: pixels [B, C, H, W] bf16 -> moments [B, 2L, H/2^(n-1), W/2^(n-1)] bf16"""
    L = _l.load()
    _chk(x, BF16, "x")
    x = x.contiguous()
    B, _, H, W = x.shape
    st = table.struct
    f = 1 << (st.n_levels - 1)
    need = L.st355_vae_encode_workspace(C.byref(st), B, H, W)
    if need == 0:
        raise _l.St355Error("vae_encode: " + L.st355_last_error().decode("utf-8", "replace"))
    ws = _vae_ws.get(x.device.index)
    if ws is None or ws.numel() < need:
        ws = _vae_ws[x.device.index] = torch.empty(need, dtype=torch.uint8, device=x.device)
    out = torch.empty(B, 2 * st.latent_channels, H // f, W // f, dtype=BF16, device=x.device)
    _l.check(L.st355_vae_encode(_stream(), C.byref(st), _ptr(x), _ptr(out), B, H, W, _ptr(ws), ws.numel()), "vae_encode")
    return out


def tokens_to_grid(tokens, B: int, H: int, W: int, residual=None):
    L = _l.load()
    _chk(tokens, BF16, "tokens")
    g = _grid_out(B, H, W, tokens.shape[1], tokens.device)
    _l.check(L.st355_tokens_to_grid(_stream(), _ptr(tokens), _ptr(residual), _ptr(g), B, H, W, tokens.shape[1]), "tokens_to_grid")
    return g


def grid_to_tokens(g, B: int, H: int, W: int):
    L = _l.load()
    _chk(g, BF16, "grid")
    t = torch.empty(B * H * W, g.shape[1], dtype=BF16, device=g.device)
    _l.check(L.st355_grid_to_tokens(_stream(), _ptr(g), _ptr(t), B, H, W, g.shape[1]), "grid_to_tokens")
    return t


_gn_ws = {}


def _gn_workspace(B, H, W, Cn, device):
    need = _l.load().st355_groupnorm_workspace(B, H, W, Cn)
    ws = _gn_ws.get(device.index)
    if ws is None or ws.numel() < need:
        ws = _gn_ws[device.index] = torch.empty(need, dtype=torch.uint8, device=device)
    return ws


def groupnorm_fwd(x, gamma, beta, B: int, H: int, W: int, groups: int = 32, eps: float = 1e-5, silu: bool = True, out_tokens: bool = False):
    """returns (y, stats).  y: grid [.., C] or dense tokens [B*H*W, C]; stats [B,C,2] fp32 for the backward"""
    L = _l.load()
    _chk(x, BF16, "x"); _chk(gamma, BF16, "gamma"); _chk(beta, BF16, "beta")
    Cn = x.shape[1]
    y = torch.empty(B * H * W, Cn, dtype=BF16, device=x.device) if out_tokens else _grid_out(B, H, W, Cn, x.device)
    stats = torch.empty(B, Cn, 2, dtype=F32, device=x.device)
    _l.check(L.st355_groupnorm_fwd(_stream(), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), B, H, W, Cn, groups, eps, 1 if silu else 0,
                                   1 if out_tokens else 0, _ptr(_gn_workspace(B, H, W, Cn, x.device))), "groupnorm_fwd")
    return y, stats


def groupnorm_bwd(dy, x, gamma, beta, stats, B: int, H: int, W: int, groups: int = 32, silu: bool = True, dy_tokens: bool = False, dadd=None,
                  dgamma=None, dbeta=None, accumulate_params: bool = False):
    L = _l.load()
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(stats, F32, "stats")
    Cn = x.shape[1]
    dx = _grid_out(B, H, W, Cn, x.device)
    _l.check(L.st355_groupnorm_bwd(_stream(), _ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(stats), _ptr(dadd), _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                   B, H, W, Cn, groups, 1 if silu else 0, 1 if dy_tokens else 0, 1 if accumulate_params else 0,
                                   _ptr(_gn_workspace(B, H, W, Cn, x.device))), "groupnorm_bwd")
    return dx


def layernorm_fwd(x, weight, bias, eps: float = 1e-5, out=None):
    L = _l.load()
    _chk(x, BF16, "x"); _chk(weight, BF16, "weight"); _chk(bias, BF16, "bias")
    rows, D = x.shape
    if out is None:
        out = torch.empty(rows, D, dtype=BF16, device=x.device)
    _l.check(L.st355_layernorm_fwd(_stream(), _ptr(x), _rows(x, "x"), _ptr(weight), _ptr(bias), _ptr(out), _rows(out, "out"), rows, D, eps), "layernorm_fwd")
    return out


def layernorm_bwd(dy, x, weight, dres=None, eps: float = 1e-5):
    L = _l.load()
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(weight, BF16, "weight")
    rows, D = x.shape
    dx = torch.empty(rows, D, dtype=BF16, device=x.device)
    _l.check(L.st355_layernorm_bwd(_stream(), _ptr(dy), _rows(dy, "dy"), _ptr(x), _rows(x, "x"), _ptr(weight), _ptr(dres),
                                   _rows(dres, "dres") if dres is not None else 0, _ptr(dx), _rows(dx, "dx"), rows, D, eps), "layernorm_bwd")
    return dx


_lnp_ws = {}


def layernorm_param_grads(dy, x, dweight, dbias, eps: float = 1e-5, accumulate: bool = False):
    L = _l.load()
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(dweight, F32, "dweight"); _chk(dbias, F32, "dbias")
    rows, D = x.shape
    need = L.st355_layernorm_param_grads_workspace(D)
    ws = _lnp_ws.get(x.device.index)
    if ws is None or ws.numel() < need:
        ws = _lnp_ws[x.device.index] = torch.empty(need, dtype=torch.uint8, device=x.device)
    _l.check(L.st355_layernorm_param_grads(_stream(), _ptr(dy), _rows(dy, "dy"), _ptr(x), _rows(x, "x"), rows, D, eps, _ptr(dweight), _ptr(dbias),
                                           1 if accumulate else 0, _ptr(ws)), "layernorm_param_grads")


def geglu_fwd(h):
    L = _l.load()
    _chk(h, BF16, "h")
    M, F2 = h.shape
    out = torch.empty(M, F2 // 2, dtype=BF16, device=h.device)
    _l.check(L.st355_geglu_fwd(_stream(), _ptr(h), _rows(h, "h"), _ptr(out), M, F2 // 2), "geglu_fwd")
    return out


def geglu_bwd(h, dout):
    L = _l.load()
    _chk(h, BF16, "h"); _chk(dout, BF16, "dout")
    M, F2 = h.shape
    dh = torch.empty(M, F2, dtype=BF16, device=h.device)
    _l.check(L.st355_geglu_bwd(_stream(), _ptr(h), _rows(h, "h"), _ptr(dout.contiguous()), _ptr(dh), _rows(dh, "dh"), M, F2 // 2), "geglu_bwd")
    return dh


def head_split(src, B: int, H: int, d: int, S: int, want_x: bool = True, want_xt: bool = True, d_src: Optional[int] = None):
    """src: 2-D column-block view [B*S, H*d_src] (row stride free) -> (X [B,H,S,d] | None, Xt [B,H,d,Sp] | None, Sp); d_src < d: zero-padded heads"""
    L = _l.load()
    _chk(src, BF16, "src")
    Sp = (S + 63) // 64 * 64
    X = torch.empty(B, H, S, d, dtype=BF16, device=src.device) if want_x else None
    Xt = None
    if want_xt:                               # only the pad columns [S, Sp) need zeros: the kernel writes every column < S (a full-size fill per attention before r6)
        Xt = torch.empty(B, H, d, Sp, dtype=BF16, device=src.device)
        if Sp > S:
            Xt[..., S:].zero_()
    _l.check(L.st355_head_split_pad(_stream(), _ptr(src), _rows(src, "src"), _ptr(X), _ptr(Xt), B, H, d if d_src is None else d_src, d, S, Sp), "head_split")
    return X, Xt, Sp


def head_merge(dX, dst, B: int, H: int, d: int, S: int, d_src: Optional[int] = None):
    L = _l.load()
    _chk(dX, BF16, "dX"); _chk(dst, BF16, "dst")
    _l.check(L.st355_head_merge_pad(_stream(), _ptr(dX), _ptr(dst), _rows(dst, "dst"), B, H, d if d_src is None else d_src, d, S), "head_merge")
    return dst


def attn_cross_fwd(Q, K, Vt, O, lse2, B, H, Sq, Sk, Skp, d, scale: float, key_bias=None, O_res=None):
    L = _l.load()
    _chk(Q, BF16, "Q"); _chk(K, BF16, "K"); _chk(Vt, BF16, "Vt"); _chk(O, BF16, "O"); _chk(lse2, F32, "lse2")
    if O_res is not None:
        _res_ok(O, O_res)
        _l.check(L.st355_attn_fwd_res(_stream(), _ptr(Q), _ptr(K), _ptr(Vt), _ptr(key_bias), _ptr(O), _rows(O, "O"), _ptr(O_res), _ptr(lse2),
                                      B, H, Sq, Sk, Skp, d, scale), "attn_fwd_res")
        return
    _l.check(L.st355_attn_cross_fwd(_stream(), _ptr(Q), _ptr(K), _ptr(Vt), _ptr(key_bias), _ptr(O), _rows(O, "O"), _ptr(lse2),
                                    B, H, Sq, Sk, Skp, d, scale), "attn_cross_fwd")


def attn_cross_bwd(Q, K, Qt, Kt, v_rows, O, dO, lse2, dQ, dK, dv_rows, B, H, Sq, Sqp, Sk, Skp, d, scale: float, key_bias=None, O_res=None):
    L = _l.load()
    need = L.st355_attn_bwd_workspace(B, H, Sq, Sqp, d)
    key = (Q.device.index,)
    ws = _attn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=Q.device)
        _attn_ws[key] = ws
    if O_res is not None:
        _res_ok(O, O_res)
        _l.check(L.st355_attn_bwd_res(_stream(), _ptr(Q), _ptr(K), _ptr(Qt), _ptr(Kt), _ptr(v_rows), _rows(v_rows, "v_rows"),
                                      _ptr(O), _rows(O, "O"), _ptr(O_res), _ptr(dO), _rows(dO, "dO"), _ptr(lse2), _ptr(key_bias),
                                      _ptr(dQ), _ptr(dK), _ptr(dv_rows), _rows(dv_rows, "dv_rows"), B, H, Sq, Sqp, Sk, Skp, d, scale, _ptr(ws)), "attn_bwd_res")
        return
    _l.check(L.st355_attn_cross_bwd(_stream(), _ptr(Q), _ptr(K), _ptr(Qt), _ptr(Kt), _ptr(v_rows), _rows(v_rows, "v_rows"),
                                    _ptr(O), _rows(O, "O"), _ptr(dO), _rows(dO, "dO"), _ptr(lse2), _ptr(key_bias),
                                    _ptr(dQ), _ptr(dK), _ptr(dv_rows), _rows(dv_rows, "dv_rows"), B, H, Sq, Sqp, Sk, Skp, d, scale, _ptr(ws)),
             "attn_cross_bwd")


def softmax_rows_(x, scale: float = 1.0):
    """in place: x[r, :] = softmax(scale * x[r, :]) over the last dim of a 2-D bf16 view"""
    L = _l.load()
    _chk(x, BF16, "x")
    _l.check(L.st355_softmax_rows(_stream(), _ptr(x), _rows(x, "x"), x.shape[0], x.shape[1], scale), "softmax_rows")
    return x


def softmax_rows_bwd_(p, dp, scale: float = 1.0):
    """in place on dp: ds = scale * p * (dp - rowsum(dp * p))"""
    L = _l.load()
    _chk(p, BF16, "p"); _chk(dp, BF16, "dp")
    if _rows(p, "p") != _rows(dp, "dp"):
        raise _l.St355Error("softmax_rows_bwd: p and dp must share a row stride")
    _l.check(L.st355_softmax_rows_bwd(_stream(), _ptr(p), _ptr(dp), _rows(dp, "dp"), p.shape[0], p.shape[1], scale), "softmax_rows_bwd")
    return dp
