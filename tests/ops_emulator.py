"""CPU emulation of the kernel CONTRACTS behind `simpletuner_amd.ops` — TEST INFRASTRUCTURE ONLY.

The engines (flux/transformer.py, ...) are host code that sequences libst355 launches: which kernel, on which operand view, into which buffer, with
which saved activation.  That sequencing is most of what can go wrong in a hand-written backward, and none of it needs a GPU to be wrong.  `install()`
monkeypatches the wrappers of `simpletuner_amd.ops` with plain-torch functions that honour the same argument contracts (st355.h): operands stay bf16
in memory, the arithmetic is fp32 with one bf16 rounding at each store (as the kernels do), outputs are written IN PLACE into the views the engine
passes (strided row blocks of joint buffers included), and the preconditions the C entry points enforce (dtypes, unit inner strides, the 64-row
contraction granule of the TN GEMM, ...) raise here too.  A `-m "not gpu"` test can then run an engine's forward + backward on the CPU and compare with
the oracle's autograd — the GPU parity tests remain the proof for the kernels themselves.

Covered: the GEMM family and its epilogues, the TN weight-gradient GEMM, token-axis reductions, AdaLN / LayerNorm / GroupNorm / RMSNorm + RoPE forward and backward,
self- and cross-attention, the grid-buffer convolution path of the UNet / VAE (conv-as-GEMM, im2col, up / down sampling), GEGLU, rank-space LoRA products, noising,
the fused losses, AdamW (+ EMA), gradient norm / clipping — 66 wrappers (`_EMULATED`).  Nothing in the product imports this module; the product has no CPU path (ops.* raise
on host tensors; tests/test_product_isolation_cpu.py).  NOT emulated: the block-level C entry points (st355_block_flux_*, st355_vae_encode), the fp8 Linears and AdamWBF16's
stochastic rounding — tests switch those paths off, exactly like the A/B env switches do.  The fused QKV projection epilogue (ST355_EPI_QK_NORM_ROPE) and the attention
backward with its RoPE / RMSNorm epilogues ARE emulated: that is Flux's default host path at head_dim 128 with tile-aligned streams.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch

from simpletuner_amd.lib import EPI_ADD, EPI_GATE_RESIDUAL, EPI_GEGLU, EPI_GEGLU_GRAD, EPI_GELU, EPI_HEADS, EPI_MUL_GELU_GRAD, EPI_NONE, EPI_QK_NORM_ROPE      # the ST355_EPI_* values (st355.h)

BF16, F32 = torch.bfloat16, torch.float32


class EmuError(RuntimeError):
    pass


def _need(cond, msg):
    if not cond:
        raise EmuError(msg)


def _chk(t, dtype, name):
    _need(torch.is_tensor(t) and t.dtype == dtype, f"{name}: expected {dtype}, got {getattr(t, 'dtype', type(t))}")


def _rows(t, name):
    _need(t.dim() == 2 and t.stride(1) == 1, f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} stride {t.stride()}")
    return t.stride(0)


def _seg(t, name):
    """the operand forms ops._seg accepts: 2-D row-major, or [segments, rows, cols] whose segment stride is a whole number of rows"""
    if t.dim() == 2:
        _rows(t, name)
        return t
    _need(t.dim() == 3 and t.stride(2) == 1 and t.stride(1) > 0 and t.stride(0) % t.stride(1) == 0, f"{name}: not a segmented row view: {tuple(t.shape)} {t.stride()}")
    return t


def _al(t, nbytes, name):
    """pointer alignment the C entry points require (host allocations are 64-byte aligned, so a view's offset decides, as on the device)"""
    _need(t.data_ptr() % nbytes == 0, f"{name}: pointer not {nbytes}-byte aligned")


def _ld(t, mult, name):
    """leading dimension (row stride in elements) of a 2-D / segmented operand must be a multiple of `mult`"""
    ld = t.stride(-2)
    _need(ld % mult == 0, f"{name}: leading dimension {ld} is not a multiple of {mult}")
    return ld


def _segcheck(t, M, name):
    if t.dim() == 3:
        _need(t.shape[1] % 256 == 0 and M % t.shape[1] == 0 and t.stride(0) // t.stride(1) >= t.shape[1], f"{name}: seg_rows {t.shape[1]} must be a multiple of 256 dividing M = {M}")


def _flat(t):
    return t.reshape(-1, t.shape[-1]).float()


def _put(dst, val):
    """store fp32 values into a (possibly strided / segmented) bf16 or fp32 destination view, in place"""
    dst.copy_(val.reshape(dst.shape).to(dst.dtype))
    return dst


def _gelu(x):
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))


def _gelu_grad(x):
    u = 0.7978845608028654 * (x + 0.044715 * x ** 3)
    t = torch.tanh(u)
    return 0.5 * (1.0 + t) + 0.5 * x * (1.0 - t * t) * 0.7978845608028654 * (1.0 + 3 * 0.044715 * x * x)


def _per_batch(v, rows_total, rows_per_batch):
    """[nb, N] per-sample rows -> [rows_total, N]"""
    nb = rows_total // rows_per_batch
    _need(nb * rows_per_batch == rows_total and v.shape[0] >= nb, f"per-batch operand: {rows_total} rows / {rows_per_batch} per batch vs {v.shape[0]} rows")
    return v[:nb].float().repeat_interleave(rows_per_batch, dim=0)


# ------------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------------
class _QkRope:
    """what ops.qk_rope hands to gemm(..., rope=...): the operands of the fused QKV epilogue (st355_qk_rope in st355.h), as tensors"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def qk_rope(Q, K, rrms, wq, wk, cos, sin, H, S, pos0, eps=1e-6, Vt=None):
    _need(tuple(cos.shape) == (S, 64) and tuple(sin.shape) == (S, 64) and cos.is_contiguous() and sin.is_contiguous(), f"qk_rope: cos / sin must be contiguous [S={S}, 64] per-pair tables")
    _chk(Q, BF16, "Q"); _chk(K, BF16, "K"); _chk(rrms, F32, "rrms"); _chk(cos, F32, "cos"); _chk(sin, F32, "sin")
    _need(H % 2 == 0 and Q.shape[1] == H and Q.shape[2] == S and Q.shape[3] == 128, "qk_rope: head_dim 128, even H")
    if Vt is not None:
        _chk(Vt, BF16, "Vt")
        _need(Vt.shape[-1] >= S and Vt.shape[-1] % 8 == 0 and pos0 % 8 == 0, "qk_rope: V^T needs Sp >= S, Sp and pos0 multiples of 8")
    return _QkRope(Q=Q, K=K, rrms=rrms, wq=wq, wk=wk, cos=cos, sin=sin, H=H, S=S, pos0=pos0, eps=eps, Vt=Vt)


def _rot_pairs(x, cos_p, sin_p):
    """rotation of the interleaved channel pairs by ONE angle per pair: cos_p / sin_p [T, d/2] broadcast over [B, T, H, d]"""
    c, s = cos_p[None, :, None, :], sin_p[None, :, None, :]
    x0, x1 = x[..., 0::2], x[..., 1::2]
    o = torch.empty_like(x)
    o[..., 0::2] = x0 * c - x1 * s
    o[..., 1::2] = x1 * c + x0 * s
    return o


def _rot_pairs_T(g, cos_p, sin_p):
    c, s = cos_p[None, :, None, :], sin_p[None, :, None, :]
    g0, g1 = g[..., 0::2], g[..., 1::2]
    o = torch.empty_like(g)
    o[..., 0::2] = g0 * c + g1 * s
    o[..., 1::2] = g1 * c - g0 * s
    return o


def _fused_qkv_epilogue(acc, out, rope, rows_per_batch):
    """ST355_EPI_QK_NORM_ROPE (st355.h): acc [M, 3D] -> roped head-major Q / K at joint positions pos0 + m % rows_per_batch, 1/rms, row-major V (+ V^T)"""
    M, N = acc.shape
    H, S, pos0 = rope.H, rope.S, rope.pos0
    D = H * 128
    _need(N == 3 * D and rows_per_batch > 0 and rows_per_batch % 256 == 0 and M % rows_per_batch == 0 and pos0 + rows_per_batch <= S,
          "gemm: EPI_QK_NORM_ROPE rows_per_batch must be a multiple of 256 dividing M, pos0 + rows_per_batch <= S, N = 3 * H * 128")
    B, R = M // rows_per_batch, rows_per_batch
    c, s = rope.cos[pos0:pos0 + R], rope.sin[pos0:pos0 + R]
    rr = rope.rrms.view(B, S, 2 * H)
    for j, (w, dst) in enumerate(((rope.wq, rope.Q), (rope.wk, rope.K))):
        x = acc[:, j * D:(j + 1) * D].reshape(B, R, H, 128)
        r = torch.rsqrt((x * x).mean(dim=-1, keepdim=True) + rope.eps) if w is not None else torch.ones(B, R, H, 1)
        y = x * r * (w.float() if w is not None else 1.0)
        dst[:, :, pos0:pos0 + R] = _rot_pairs(y, c, s).permute(0, 2, 1, 3).to(BF16)
        rr[:, pos0:pos0 + R, j * H:(j + 1) * H] = r[..., 0]
    v = acc[:, 2 * D:]
    _put(out, v)
    if rope.Vt is not None:
        rope.Vt[:, :, :, pos0:pos0 + R] = v.reshape(B, R, H, 128).permute(0, 2, 3, 1).to(BF16)
    return out


class _Heads:
    """what ops.heads hands to gemm(..., heads=...): the destinations of the head-splitting epilogue (st355_heads in st355.h), as tensors"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def heads(Q, K, Vt, H, S, pos0, n_q, n_k):
    for t, nm in ((Q, "Q"), (K, "K"), (Vt, "Vt")):
        if t is not None:
            _chk(t, BF16, nm); _al(t, 16, nm)
            _need(t.is_contiguous(), f"heads: {nm} must be contiguous")
    _need(n_q % 64 == 0 and n_k % 64 == 0 and (n_q == 0 or (Q is not None and n_q == H * 64)) and (n_k == 0 or (K is not None and n_k == H * 64)),
          "heads: every present part is H heads of 64 columns")
    for t in (Q, K):
        _need(t is None or (t.shape[1] == H and t.shape[2] == S and t.shape[3] == 64), "heads: Q / K are [B, H, S, 64]")
    if Vt is not None:
        _need(Vt.shape[1] == H and Vt.shape[2] == 64 and Vt.shape[3] >= S and Vt.shape[3] % 8 == 0 and pos0 % 8 == 0, "heads: V^T is [B, H, 64, Sp], Sp >= S, Sp and pos0 multiples of 8")
    return _Heads(Q=Q, K=K, Vt=Vt, H=H, S=S, pos0=pos0, n_q=n_q, n_k=n_k)


def _heads_epilogue(acc, out, h, rows_per_batch):
    """ST355_EPI_HEADS (st355.h): acc [M, n_q + n_k + n_v] -> head-major Q / K at positions pos0 + m % rows_per_batch, row-major V rows (+ head-major V^T)"""
    M, N = acc.shape
    n_v = N - h.n_q - h.n_k
    _need(N % 64 == 0 and (n_v == 0 or n_v == h.H * 64), "gemm: EPI_HEADS: the v part is H heads of 64 columns")
    _need(rows_per_batch > 0 and M % rows_per_batch == 0 and h.pos0 >= 0 and h.pos0 + rows_per_batch <= h.S, "gemm: EPI_HEADS rows_per_batch / pos0 / S")
    B, R, p0 = M // rows_per_batch, rows_per_batch, h.pos0
    col = 0
    for n, dst in ((h.n_q, h.Q), (h.n_k, h.K)):
        if n:
            dst[:B, :, p0:p0 + R] = acc[:, col:col + n].reshape(B, R, h.H, 64).permute(0, 2, 1, 3).to(BF16)
            col += n
    if n_v:
        _need(out is not None and tuple(out.shape) == (M, n_v), f"gemm: EPI_HEADS out is the row-major V [{M}, {n_v}]")
        v = acc[:, col:]
        _put(out, v)
        if h.Vt is not None:
            _need(R % 8 == 0, "gemm: EPI_HEADS V^T needs rows_per_batch % 8 == 0")
            h.Vt[:B, :, :, p0:p0 + R] = v.reshape(B, R, h.H, 64).permute(0, 2, 3, 1).to(BF16)
    else:
        _need(h.Vt is None, "gemm: EPI_HEADS V^T without v heads")
    return out


def gemm(a, w, bias=None, out=None, epilogue=EPI_NONE, a2=None, b2=None, aux_out=None, aux_in=None, gate=None, rows_per_batch=0, k2_real=0, rope=None, heads=None):
    _chk(a, BF16, "a"); _chk(w, BF16, "w")
    _seg(a, "a"); _rows(w, "w")
    _need(epilogue != EPI_QK_NORM_ROPE or (rope is not None and out is not None), "gemm: EPI_QK_NORM_ROPE needs rope=qk_rope(...) and out= (the V destination)")
    A = _flat(a)
    M, K = A.shape
    N = w.shape[0]
    _need(w.shape[1] == K, f"gemm: K mismatch {K} vs {w.shape[1]}")
    _need(K % 64 == 0, f"gemm: K = {K} is not a multiple of 64")
    _need(N % 4 == 0, f"gemm: N = {N} must be a multiple of 4")
    _al(a, 16, "a"); _al(w, 16, "w"); lda = _ld(a, 8, "a"); ldb = _ld(w, 8, "w"); _segcheck(a, M, "a")
    _need(256 * max(lda, ldb) * 2 + K * 2 < (1 << 31), "gemm: a 256-row tile must fit 32-bit buffer offsets")
    acc = A @ w.float().t()
    if a2 is not None:
        _chk(a2, BF16, "a2"); _chk(b2, BF16, "b2"); _seg(a2, "a2"); _rows(b2, "b2")
        _al(a2, 16, "a2"); _al(b2, 16, "b2"); _ld(a2, 8, "a2"); _ld(b2, 8, "b2"); _segcheck(a2, M, "a2")
        A2 = _flat(a2)
        _need(A2.shape[0] == M and b2.shape == (N, A2.shape[1]) and A2.shape[1] % 64 == 0, "gemm: low-rank / second-segment shape mismatch")
        acc = acc + A2 @ b2.float().t()
    if bias is not None:
        _chk(bias, BF16, "bias")
        _need(bias.is_contiguous() and bias.numel() == N, "gemm: bias")
        acc = acc + bias.float()
    if epilogue == EPI_QK_NORM_ROPE:
        _chk(out, BF16, "out"); _seg(out, "out"); _al(out, 16, "out"); _ld(out, 8, "out"); _segcheck(out, M, "out")
        _need(out.numel() == M * (N // 3) and out.shape[-1] == N // 3, f"gemm: the V destination is {tuple(out.shape)}, expected {M}x{N // 3}")
        _need(out.dim() == 2 or out.shape[1] == rows_per_batch, "gemm: EPI_QK_NORM_ROPE segments are the per-sample row blocks (seg_rows == rows_per_batch)")
        return _fused_qkv_epilogue(acc, out, rope, rows_per_batch)
    if epilogue == EPI_HEADS:
        _need(heads is not None and a.dim() == 2, "gemm: EPI_HEADS needs heads=heads(...) and a plain (unsegmented) problem")
        n_v = N - heads.n_q - heads.n_k
        if n_v:
            if out is None:
                out = torch.empty(M, n_v, dtype=BF16, device=a.device)
            _chk(out, BF16, "out"); _al(out, 16, "out"); _ld(out, 8, "out")
        else:
            _need(out is None, "gemm: EPI_HEADS without v heads takes no out=")
        return _heads_epilogue(acc, out, heads, rows_per_batch)
    if epilogue in (EPI_GEGLU, EPI_GEGLU_GRAD):
        # the UNet feed-forward's GEGLU inside its two GEMMs (st355.h): interleaved columns — every 64 = [32 values | the 32 gates of the same features]
        _need(N % 64 == 0 and a2 is None and gate is None and a.dim() == 2, "gemm: the GEGLU epilogues take a plain problem with N % 64 == 0")
        erf_gelu = lambda t: torch.nn.functional.gelu(t)
        if epilogue == EPI_GEGLU:
            _need(aux_out is not None and tuple(aux_out.shape) == (M, N), "gemm: EPI_GEGLU keeps the interleaved pre-activation in aux_out [M, N]")
            _chk(aux_out, BF16, "aux_out"); _al(aux_out, 16, "aux_out"); _ld(aux_out, 8, "aux_out")
            pre = acc.to(BF16)
            aux_out.copy_(pre)
            pv = pre.float().view(M, N // 64, 2, 32)
            val = (pv[:, :, 0] * erf_gelu(pv[:, :, 1])).reshape(M, N // 2)
            if out is None:
                out = torch.empty(M, N // 2, dtype=BF16, device=a.device)
            _need(tuple(out.shape) == (M, N // 2), f"gemm: EPI_GEGLU out is {tuple(out.shape)}, expected {M}x{N // 2}")
        else:
            _need(bias is None and aux_in is not None and tuple(aux_in.shape) == (M, 2 * N), "gemm: EPI_GEGLU_GRAD reads the interleaved pre-activation from aux_in [M, 2N]; no bias")
            _chk(aux_in, BF16, "aux_in"); _al(aux_in, 16, "aux_in"); _ld(aux_in, 8, "aux_in")
            pv = aux_in.float().view(M, N // 32, 2, 32)
            d = acc.to(BF16).float().view(M, N // 32, 32)
            gt = pv[:, :, 1].double()
            dgelu = (0.5 * (1 + torch.erf(gt / math.sqrt(2.0))) + gt * torch.exp(-0.5 * gt * gt) / math.sqrt(2.0 * math.pi)).float()
            val = torch.stack([d * erf_gelu(pv[:, :, 1]), d * pv[:, :, 0] * dgelu], dim=2).reshape(M, 2 * N)
            if out is None:
                out = torch.empty(M, 2 * N, dtype=BF16, device=a.device)
            _need(tuple(out.shape) == (M, 2 * N), f"gemm: EPI_GEGLU_GRAD out is {tuple(out.shape)}, expected {M}x{2 * N}")
        _chk(out, BF16, "out"); _al(out, 16, "out"); _ld(out, 8, "out")
        return _put(out, val)
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    _chk(out, BF16, "out"); _seg(out, "out")
    _need(out.numel() == M * N and out.shape[-1] == N, f"gemm: out is {tuple(out.shape)}, expected {M}x{N}")
    _al(out, 8, "out"); _ld(out, 4, "out"); _segcheck(out, M, "out")
    if aux_in is not None:
        _chk(aux_in, BF16, "aux_in"); _seg(aux_in, "aux_in")
        _need(aux_in.numel() == M * N, "gemm: aux_in shape")
        _ld(aux_in, 4, "aux_in"); _segcheck(aux_in, M, "aux_in")
    if aux_out is not None:
        _chk(aux_out, BF16, "aux_out"); _seg(aux_out, "aux_out")
        _need(aux_out.numel() == M * N, "gemm: aux_out shape")
        _ld(aux_out, 4, "aux_out"); _segcheck(aux_out, M, "aux_out")
    if epilogue == EPI_NONE:
        val = acc
    elif epilogue == EPI_ADD:
        _need(aux_in is not None, "gemm: EPI_ADD needs aux_in")
        val = acc + _flat(aux_in)
    elif epilogue == EPI_GELU:
        if aux_out is not None:
            _put(aux_out, acc)                      # the pre-activation, for the backward
        val = _gelu(acc)
    elif epilogue == EPI_MUL_GELU_GRAD:
        _need(aux_in is not None, "gemm: EPI_MUL_GELU_GRAD needs aux_in (the saved pre-activation)")
        val = acc * _gelu_grad(_flat(aux_in))
    elif epilogue == EPI_GATE_RESIDUAL:
        _need(aux_in is not None and gate is not None and rows_per_batch > 0, "gemm: EPI_GATE_RESIDUAL needs aux_in, gate, rows_per_batch")
        _chk(gate, BF16, "gate"); _rows(gate, "gate")
        if aux_out is not None:
            _put(aux_out, acc)                      # the un-gated branch output
        val = _flat(aux_in) + _per_batch(gate, M, rows_per_batch) * acc
    else:
        raise EmuError(f"gemm: epilogue {epilogue}")
    return _put(out, val)


def gemm_grouped(problems):
    outs = []
    for pr in problems:
        pr = dict(pr)
        outs.append(gemm(pr.pop("a"), pr.pop("w"), **pr))
    return outs


def gemm_tn(Lm, R, out=None, accumulate=False):
    segs = []
    for t in (Lm, R):               # a 3-D [B, rows, C] strided view = the segmented-contraction form (st355_gemm_tn_seg_bf16): rows % 64 == 0, >= 128
        if t.dim() == 3:
            _need(t.stride(2) == 1 and t.stride(0) % t.stride(1) == 0 and t.shape[1] % 64 == 0 and t.shape[1] >= 128, f"gemm_tn: bad segmented operand {tuple(t.shape)} {t.stride()}")
            segs.append(t.shape[1])
    _need(len(set(segs)) <= 1, "gemm_tn: two segmented operands must share the segment length")
    if segs:
        Lm = Lm.reshape(-1, Lm.shape[-1]) if Lm.dim() == 3 else Lm
        R = R.reshape(-1, R.shape[-1]) if R.dim() == 3 else R
        _need(Lm.shape[0] % segs[0] == 0, "gemm_tn: the segment length must divide the contraction length")
    _chk(Lm, BF16, "L"); _chk(R, BF16, "R"); _rows(Lm, "L"); _rows(R, "R")
    M, P = Lm.shape
    _need(R.shape[0] == M, "gemm_tn: operands must share the contraction length")
    _need(M % 64 == 0, f"gemm_tn: contraction length {M} is not a multiple of 64 (zero-pad the rows)")
    _need(P % 8 == 0 and R.shape[1] % 8 == 0, "gemm_tn: P, Q must be multiples of 8")
    _al(Lm, 16, "L"); _al(R, 16, "R")
    _need(M * max(_ld(Lm, 8, "L"), _ld(R, 8, "R")) * 2 < (1 << 31), "gemm_tn: operand too large for the 32-bit buffer offsets (2 GiB)")
    val = Lm.float().t() @ R.float()
    if out is None:
        _need(not accumulate, "gemm_tn: accumulate needs an output tensor")
        out = torch.empty(P, R.shape[1], dtype=BF16, device=Lm.device)
    _chk(out, BF16, "out"); _rows(out, "out")
    _need(tuple(out.shape) == (P, R.shape[1]), f"gemm_tn: out is {tuple(out.shape)}, expected {(P, R.shape[1])}")
    _al(out, 16, "out"); _ld(out, 8, "out")
    if accumulate:
        val = val + out.float()
    return _put(out, val)


def colsum_prod(a, out, b=None, rows_per_batch=None, mode=0, prev=None, shift=None, scale=None, accumulate=False):
    _chk(a, BF16, "a"); _chk(out, F32, "out"); _rows(a, "a"); _rows(out, "out")
    rows, N = a.shape
    rpb = rows if rows_per_batch is None else rows_per_batch
    nb = rows // rpb
    _need(nb * rpb == rows and out.shape[0] >= nb and out.shape[1] == N, f"colsum_prod: {rows} rows, {rpb} per batch, out {tuple(out.shape)}")
    _need(N % 8 == 0, "colsum_prod: N must be a multiple of 8"); _al(a, 16, "a"); _ld(a, 8, "a")
    prod = a.float()
    if b is not None:
        _chk(b, BF16, "b"); _rows(b, "b"); _al(b, 16, "b"); _ld(b, 8, "b")
        _need(tuple(b.shape) == (rows, N), "colsum_prod: b shape")
        prod = prod * b.float()
    s = prod.view(nb, rpb, N).sum(dim=1)
    if mode == 1:
        _chk(shift, BF16, "shift"); _chk(scale, BF16, "scale"); _chk(prev, F32, "prev")
        _need(_rows(shift, "shift") == _rows(scale, "scale"), "colsum_prod: shift and scale must share a row stride")
        sc = 1.0 + scale[:nb].float()
        sc = torch.where(sc.abs() > 1e-6, sc, torch.where(sc < 0, torch.full_like(sc, -1e-6), torch.full_like(sc, 1e-6)))      # k_colsum_finalize's guard
        s = (s - shift[:nb].float() * prev[:nb]) / sc
    if accumulate:
        s = s + out[:nb]
    out[:nb] = s
    return out


def _stat_put(dst, val, reduce_batches, accumulate=False):
    """st355_stat_out semantics: per-batch fp32 rows, or one row over the batches (fp32 / bf16); overwrite unless accumulate"""
    if dst is None:
        return
    v = val.sum(dim=0) if reduce_batches else val
    _need(dst.dtype == F32 or (reduce_batches and dst.dtype == BF16), "stat_out: per-batch sums are fp32; a bf16 destination is one row over the batches")
    _need(tuple(dst.shape) == tuple(v.shape), f"stat_out: destination {tuple(dst.shape)}, sums {tuple(v.shape)}")
    dst.copy_((dst.float() + v if accumulate else v).to(dst.dtype))


def ln_modulate_bwd_stats(dy, x, scale, rows_per_batch, d_shift, d_scale, dres=None, gate=None, y_branch=None, d_gate=None, d_bias=None, eps=1e-6, want_gated=False,
                          out=None):
    """st355_ln_modulate_bwd_stats: ln_modulate_bwd + d shift = sum dy, d scale = sum dy * LN(x) (fp32 LN), d gate = sum dx * y (dx as stored), d bias = sum dxg"""
    dx, dxg = ln_modulate_bwd(dy, x, scale, rows_per_batch, dres=dres, gate=gate, eps=eps, want_gated=want_gated, out=out)
    rows, D = x.shape
    nb = rows // rows_per_batch
    _need(nb * rows_per_batch == rows and D <= 3072, "ln_modulate_bwd_stats: rows % rows_per_batch, D <= 3072")
    pb = lambda t: t.view(nb, rows_per_batch, D).sum(dim=1)
    xh = torch.nn.functional.layer_norm(x.float(), (D,), eps=eps)
    _stat_put(d_shift, pb(dy.float()), False); _stat_put(d_scale, pb(dy.float() * xh), False)
    if d_gate is not None:
        _need(y_branch is not None, "ln_modulate_bwd_stats: the gate gradient needs the branch output")
        _stat_put(d_gate, pb(dx.float() * y_branch.float()), False)
    if d_bias is not None:
        _need(dxg is not None, "ln_modulate_bwd_stats: the bias gradient is the column sum of the gated output")
        _stat_put(d_bias, pb(dxg.float()), True)
    return dx, dxg


def scale_cols_stats(x, gate, rows_per_batch, y_branch=None, d_gate=None, d_bias=None, out=None):
    g = scale_cols(x, gate, rows_per_batch, out=out)
    M, N = x.shape
    nb = M // rows_per_batch
    pb = lambda t: t.view(nb, rows_per_batch, N).sum(dim=1)
    if d_gate is not None:
        _need(y_branch is not None, "scale_cols_stats: the gate gradient needs the branch output")
        _stat_put(d_gate, pb(x.float() * y_branch.float()), False)
    _stat_put(d_bias, pb(g.float()), True)
    return g


def colsum_rows(a, rows_per_batch, batch_stride_rows, nb, out, per_batch=False, accumulate=False):
    _chk(a, BF16, "a"); _al(a, 16, "a"); _ld(a, 8, "a")
    _need(batch_stride_rows >= rows_per_batch and a.shape[0] >= (nb - 1) * batch_stride_rows + rows_per_batch, "colsum_rows: the row blocks leave the tensor")
    v = torch.stack([a[b * batch_stride_rows:b * batch_stride_rows + rows_per_batch].float().sum(dim=0) for b in range(nb)])
    _stat_put(out, v, not per_batch, accumulate)
    return out


def transpose(src, out=None):
    _chk(src, BF16, "src"); _rows(src, "src")
    R, Cn = src.shape
    _need(R % 8 == 0 and Cn % 8 == 0, f"transpose: {R} x {Cn} (rows, cols multiples of 8)")
    if out is None:
        out = torch.empty(Cn, R, dtype=BF16, device=src.device)
    _chk(out, BF16, "out"); _rows(out, "out")
    _need(tuple(out.shape) == (Cn, R), "transpose: out shape")
    out.copy_(src.t())
    return out


def skinny_tn(Lm, R, out, so_p, so_r, r_used, alpha=1.0, accumulate=False):
    _chk(Lm, BF16, "L"); _chk(R, BF16, "R"); _chk(out, F32, "out")
    _seg(Lm, "L"); _seg(R, "R")
    L2, R2 = _flat(Lm), _flat(R)
    _need(L2.shape[0] == R2.shape[0] and R2.shape[1] in (32, 64) and r_used <= R2.shape[1], f"skinny_tn: L {tuple(L2.shape)}, R {tuple(R2.shape)}, r_used {r_used}")
    val = alpha * (L2.t() @ R2[:, :r_used])
    dst = torch.as_strided(out, (L2.shape[1], r_used), (so_p, so_r))
    dst.copy_(val + dst if accumulate else val)
    return out


def skinny_tn_multi(Lm, R, outs, so_p, so_r, r_used, alpha=1.0, accumulate=False):
    L2, R2 = _flat(_seg(Lm, "L")), _flat(_seg(R, "R"))
    _need(L2.shape[0] == R2.shape[0] and R2.shape[1] >= 128 and 1 <= len(outs) <= 4 and r_used <= 32, "skinny_tn_multi: shapes")
    for g, o in enumerate(outs):
        _chk(o, F32, "out")
        val = alpha * (L2.t() @ R2[:, 32 * g:32 * g + r_used])
        dst = torch.as_strided(o, (L2.shape[1], r_used), (so_p, so_r))
        dst.copy_(val + dst if accumulate else val)
    return outs


def lora_pack(A, Bm, scale, A_cat, A_cat_T, B_blk, B_blk_T, k2_off=0, n_off=0):
    _chk(A, F32, "A"); _chk(Bm, F32, "B")
    r, K = A.shape
    N = Bm.shape[0]
    K2, N_total = A_cat.shape[0], B_blk.shape[0]
    _need(A_cat.shape == (K2, K) and A_cat_T.shape == (K, K2) and B_blk.shape == (N_total, K2) and B_blk_T.shape == (K2, N_total), "lora_pack: operand shapes inconsistent")
    A_cat[k2_off:k2_off + r] = A.to(BF16)
    A_cat_T[:, k2_off:k2_off + r] = A.t().to(BF16)
    B_blk[n_off:n_off + N, k2_off:k2_off + r] = (scale * Bm).to(BF16)
    B_blk_T[k2_off:k2_off + r, n_off:n_off + N] = (scale * Bm).t().to(BF16)


# ------------------------------------------------------------------------------------------------
# streaming ops
# ------------------------------------------------------------------------------------------------
def flow_noise_mix(x, sigma, noise=None, seed=0, offset=0, want_target=True):
    """x_t = (1 - sigma) x + sigma n ; target = n - x.  noise=None: the kernel draws Philox normals from (seed, offset) — here torch's generator seeded the same way
    (a different stream: distributional equivalence only, as DESIGN.md states for the product vs torch.randn)"""
    _chk(x, BF16, "x"); _chk(sigma, F32, "sigma")
    B = x.shape[0]
    _need((x.numel() // B) % 8 == 0, "flow_noise_mix: per_sample must be a multiple of 8")
    if noise is None:
        noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(int(seed) * 1_000_003 + int(offset))).to(BF16)
    else:
        _chk(noise, BF16, "noise")
    s = sigma.float().view(B, *([1] * (x.dim() - 1)))
    x_t = ((1.0 - s) * x.float() + s * noise.float()).to(BF16)
    target = (noise.float() - x.float()).to(BF16) if want_target else None
    return x_t, target, noise


def ddpm_noise_mix(x, noise, sqrt_acp, sqrt_1macp, want_v=True):
    _chk(x, BF16, "x"); _chk(noise, BF16, "noise"); _chk(sqrt_acp, F32, "sqrt_acp"); _chk(sqrt_1macp, F32, "sqrt_1macp")
    B = x.shape[0]
    a, b = (t.float().view(B, *([1] * (x.dim() - 1))) for t in (sqrt_acp, sqrt_1macp))
    x_t = (a * x.float() + b * noise.float()).to(BF16)
    return x_t, ((a * noise.float() - b * x.float()).to(BF16) if want_v else None)


def flux_pack(latents):
    return patchify(latents, order=0)


def flux_unpack(packed, Cc, H, W):
    return unpatchify(packed, Cc, H, W, order=0)


def _loss_common(pred, target, weight, emask, want_grad, grad_scale, elem, delem):
    """per-element loss `elem(d)` / derivative `delem(d)` -> [* emask] -> per-sample mean (* w_b) -> batch mean; dpred = grad_scale * d loss / d pred"""
    _chk(pred, BF16, "pred"); _chk(target, BF16, "target")
    B = pred.shape[0]
    per = pred.numel() // B
    _need(per % 8 == 0, "loss: per-sample size must be a multiple of 8")
    d = pred.float().reshape(B, per) - target.float().reshape(B, per)
    mk = torch.ones(B, per)
    if emask is not None:
        _chk(emask, F32, "emask")
        em = emask.reshape(B, -1)
        mk = em.repeat(1, per // em.shape[1])
    w = torch.ones(B) if weight is None else weight.float()
    per_sample = (elem(d) * mk).sum(dim=1) * w / per
    loss = per_sample.mean().reshape(1)
    dpred = (grad_scale * (delem(d) * mk) * w[:, None] / (per * B)).reshape(pred.shape).to(BF16) if want_grad else None
    return loss, per_sample, dpred


def mse_loss(pred, target, weight=None, want_grad=True, grad_scale=1.0):
    return _loss_common(pred, target, weight, None, want_grad, grad_scale, lambda d: d * d, lambda d: 2.0 * d)


def cond_loss(pred, target, loss_type="l2", huber_c=0.1, weight=None, want_grad=True, grad_scale=1.0, emask=None):
    B = pred.shape[0]
    if loss_type == "l2":
        return _loss_common(pred, target, weight, emask, want_grad, grad_scale, lambda d: d * d, lambda d: 2.0 * d)
    c = (huber_c if torch.is_tensor(huber_c) else torch.full((B,), float(huber_c))).float().reshape(B, 1)
    k = 2.0 * c if loss_type == "huber" else torch.full_like(c, 2.0)
    return _loss_common(pred, target, weight, emask, want_grad, grad_scale, lambda d: k * (torch.sqrt(d * d + c * c) - c), lambda d: k * d / torch.sqrt(d * d + c * c))


def adamw_ema_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, grad_scale=1.0, ema=None, ema_decay=0.0, p_bf16=None):
    """torch.optim.AdamW's single-tensor order of operations over a flat arena (fp32 p / g, or bf16 p / g with fp32 moments), optional fused EMA"""
    _need(step >= 1 and p.dim() == 1 and p.is_contiguous() and g.is_contiguous(), "adamw_ema_step: flat contiguous arenas, step >= 1")
    _need(p.dtype in (F32, BF16) and g.dtype == p.dtype and m.dtype == F32 and v.dtype == F32, "adamw_ema_step: dtypes")
    _need(p.dtype == F32 or p.numel() % 8 == 0, "adamw_ema_step_bf16: n must be a multiple of 8")
    gf = g.float() * grad_scale
    pf = p.float() * (1.0 - lr * weight_decay)
    m.add_((gf - m) * (1.0 - beta1))
    v.mul_(beta2).add_((1.0 - beta2) * gf * gf)
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    pf = pf - (lr / bc1) * (m / (v.sqrt() / math.sqrt(bc2) + eps))
    p.copy_(pf.to(p.dtype))
    if ema is not None:
        ema.copy_((ema.float() - (1.0 - ema_decay) * (ema.float() - p.float()).to(ema.dtype).float()).to(ema.dtype))      # (s - p) materialised in the shadow dtype, as k_ema
    if p_bf16 is not None:
        p_bf16.copy_(pf.to(BF16))


def adamw_bf16_sr_step(p, g, m, v, shift, step, lr, beta1, beta2, eps, seg_end=None, seg_decay=None, rand_bits=None, seed=0, offset=0, grad_scale=1.0):
    """st355_adamw_bf16_sr_step's contract over flat bf16 arenas through the oracle's restatement of the reference update (oracle.train_math.adamw_bf16_step);
    the stochastic-rounding draws are host-drawn here (the kernel's Philox stream is a GPU matter: tests/test_adamw_bf16_gpu.py pins it bit for bit)"""
    from oracle import train_math as TM
    for t, nm in ((p, "p"), (g, "g"), (m, "exp_avg"), (v, "exp_avg_sq"), (shift, "shift")):
        _need(t.dtype == BF16 and t.is_contiguous() and t.numel() == p.numel(), f"adamw_bf16_sr_step: {nm} must be a contiguous bf16 arena of the same length")
    n = p.numel()
    ends = [n] if seg_end is None else [int(e) for e in seg_end]
    decs = [0.0] * len(ends) if seg_decay is None else [float(d) for d in seg_decay]
    gen = torch.Generator().manual_seed(int(seed) * 7919 + int(offset) % 1000003)
    gs = (g.float() * grad_scale).to(BF16)
    lo = 0
    for hi, dec in zip(ends, decs):
        k = hi - lo
        draws = [rand_bits[j, lo:hi] for j in range(4)] if rand_bits is not None else [torch.randint(0, 1 << 16, (k,), generator=gen, dtype=torch.int32) for _ in range(4)]
        o = TM.adamw_bf16_step(p[lo:hi], gs[lo:hi], m[lo:hi], v[lo:hi], shift[lo:hi], int(step), lr, beta1, beta2, eps, dec, draws)
        for dst, src in zip((p, m, v, shift), o):
            dst[lo:hi].copy_(src)
        lo = hi


def ema_update(shadow, param, decay):
    """s -= (1 - d) (s - p), the difference materialised in the parameter dtype (ema.py:423)"""
    _need(shadow.dtype == param.dtype, "ema_update: dtype mismatch")
    diff = (shadow.float() - param.float()).to(shadow.dtype)
    shadow.copy_((shadow.float() - (1.0 - decay) * diff.float()).to(shadow.dtype))


def grad_norm(g):
    return torch.stack([(g.float() ** 2).sum(), g.float().abs().max()])


def grad_clamp_(g, c):
    _need(g.dtype in (F32, BF16) and g.is_contiguous(), "grad_clamp: expected a contiguous fp32 / bf16 tensor")
    g.clamp_(-c, c)
    return g


def grad_clip_norm_(g, stats, max_norm, pre_scale=1.0):
    _chk(stats, F32, "stats")
    coef = min(max_norm / (float(stats[0].sqrt()) * pre_scale + 1e-6), 1.0)
    if coef < 1.0:
        g.copy_((g.float() * coef).to(g.dtype))
    return g


def timestep_proj(t, dim, scale=1.0):
    _chk(t, F32, "t")
    half = dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=F32) / half)
    a = t[:, None] * scale * f[None]
    return torch.cat([a.cos(), a.sin()], dim=1).to(BF16)


def patchify(latents, order=0):
    """[B,C,H,W] -> [B,(H/2)(W/2),4C]; order 0 = (c,dh,dw) features (Flux pack / PatchEmbed im2col), 1 = (dh,dw,c) (SD3 / PixArt proj_out layout)"""
    _chk(latents, BF16, "latents")
    B, Cn, H, W = latents.shape
    x = latents.reshape(B, Cn, H // 2, 2, W // 2, 2)
    x = x.permute(0, 2, 4, 1, 3, 5) if order == 0 else x.permute(0, 2, 4, 3, 5, 1)
    return x.reshape(B, (H // 2) * (W // 2), 4 * Cn).contiguous()


def unpatchify(packed, Cc, H, W, order=0):
    _chk(packed, BF16, "packed")
    B = packed.shape[0]
    if order == 0:
        x = packed.reshape(B, H // 2, W // 2, Cc, 2, 2).permute(0, 3, 1, 4, 2, 5)
    else:
        x = packed.reshape(B, H // 2, W // 2, 2, 2, Cc).permute(0, 5, 1, 3, 2, 4)
    return x.reshape(B, Cc, H, W).contiguous()


def silu(x):
    _chk(x, BF16, "x")
    f = x.float()
    return (f * torch.sigmoid(f)).to(BF16)


def silu_bwd(x, dy):
    _chk(x, BF16, "x"); _chk(dy, BF16, "dy")
    f = x.float()
    sg = torch.sigmoid(f)
    return (dy.float() * sg * (1.0 + f * (1.0 - sg))).to(BF16)


def add(a, b):
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    return (a.float() + b.float()).to(BF16)


def scale_cols(x, gate, rows_per_batch, out=None):
    _chk(x, BF16, "x"); _chk(gate, BF16, "gate"); _rows(x, "x"); _rows(gate, "gate")
    M, N = x.shape
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=x.device)
    return _put(out, x.float() * _per_batch(gate, M, rows_per_batch))


def gather_rows(x, idx, out=None):
    _need(x.dtype == BF16 and x.dim() == 3 and idx.dtype == torch.int32 and idx.is_contiguous(), "gather_rows: x [B, S, D] bf16, idx [B, K] contiguous int32")
    val = torch.gather(x, 1, idx.long()[:, :, None].expand(-1, -1, x.shape[2]))
    if out is None:
        return val.contiguous()
    out.copy_(val)
    return out


def scatter_rows(src, idx, dst):
    _need(src.dtype == BF16 and dst.dtype == BF16 and src.dim() == 3 and dst.dim() == 3 and idx.dtype == torch.int32, "scatter_rows: operands")
    dst.scatter_(1, idx.long()[:, :, None].expand(-1, -1, src.shape[2]), src)
    return dst


# ------------------------------------------------------------------------------------------------
# AdaLN / RMSNorm + RoPE
# ------------------------------------------------------------------------------------------------
def _ln_stats(x, eps):
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    return (x - mu) * rstd, rstd


def ln_modulate_fwd(x, scale, shift, rows_per_batch, eps=1e-6, out=None):
    _chk(x, BF16, "x"); _chk(scale, BF16, "scale"); _chk(shift, BF16, "shift")
    _rows(x, "x")
    _need(_rows(scale, "scale") == _rows(shift, "shift"), "ln_modulate_fwd: scale and shift must share a row stride")
    rows, D = x.shape
    _need(D % 8 == 0 and D <= 4096, "ln_modulate_fwd: D"); _ld(x, 8, "x"); _ld(scale, 8, "scale")
    xhat, _ = _ln_stats(x.float(), eps)
    val = xhat * (1.0 + _per_batch(scale, rows, rows_per_batch)) + _per_batch(shift, rows, rows_per_batch)
    if out is None:
        out = torch.empty(rows, D, dtype=BF16, device=x.device)
    _rows(out, "out")
    return _put(out, val)


def layer_norm_xhat(x, eps=1e-6, out=None):
    z = torch.zeros(1, x.shape[1], dtype=BF16)
    return ln_modulate_fwd(x, z, z, x.shape[0], eps=eps, out=out)


def ln_modulate_bwd(dy, x, scale, rows_per_batch, dres=None, gate=None, eps=1e-6, want_gated=False, out=None):
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(scale, BF16, "scale")
    _rows(dy, "dy"); _rows(x, "x"); _rows(scale, "scale")
    rows, D = x.shape
    _need(tuple(dy.shape) == (rows, D), "ln_modulate_bwd: dy shape")
    _need(D % 8 == 0 and D <= 4096, "ln_modulate_bwd: D"); _ld(x, 8, "x"); _ld(dy, 8, "dy"); _ld(scale, 8, "scale")
    xhat, rstd = _ln_stats(x.float(), eps)
    g = dy.float() * (1.0 + _per_batch(scale, rows, rows_per_batch))
    dx = rstd * (g - g.mean(dim=1, keepdim=True) - xhat * (g * xhat).mean(dim=1, keepdim=True))
    if dres is not None:
        _chk(dres, BF16, "dres"); _rows(dres, "dres")
        _need(tuple(dres.shape) == (rows, D), "ln_modulate_bwd: dres shape")
        dx = dx + dres.float()
    dst = torch.empty(rows, D, dtype=BF16, device=x.device) if out is None else out
    _chk(dst, BF16, "out"); _rows(dst, "out")
    _need(tuple(dst.shape) == (rows, D), "ln_modulate_bwd: out shape mismatch")
    _put(dst, dx)
    dxg = None
    if want_gated:
        _chk(gate, BF16, "gate"); _rows(gate, "gate")
        dxg = (dx * _per_batch(gate, rows, rows_per_batch)).to(BF16)
    return dst, dxg


def _rot(x, cos, sin):
    """o0 = x0 c0 - x1 s0 ; o1 = x1 c1 + x0 s1 on interleaved pairs; cos / sin [T, d] broadcast over [B, T, H, d]"""
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    x0, x1 = x[..., 0::2], x[..., 1::2]
    o = torch.empty_like(x)
    o[..., 0::2] = x0 * c[..., 0::2] - x1 * s[..., 0::2]
    o[..., 1::2] = x1 * c[..., 1::2] + x0 * s[..., 1::2]
    return o


def _rot_T(g, cos, sin):
    """the transpose of _rot: dy0 = g0 c0 + g1 s1 ; dy1 = g1 c1 - g0 s0"""
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    g0, g1 = g[..., 0::2], g[..., 1::2]
    o = torch.empty_like(g)
    o[..., 0::2] = g0 * c[..., 0::2] + g1 * s[..., 1::2]
    o[..., 1::2] = g1 * c[..., 1::2] - g0 * s[..., 0::2]
    return o


def _stream_rows(qkv, B, S, pos0, S_part):
    ld = _rows(qkv, "qkv")
    _need(qkv.shape[0] >= B * S, f"qkv: {qkv.shape[0]} rows for B * S = {B * S}")
    return torch.as_strided(qkv, (B, S_part, qkv.shape[1]), (S * ld, ld, 1), qkv.storage_offset() + pos0 * ld)


def qk_norm_rope_fwd(qkv, wq, wk, cos, sin, Q, K, Qt, Kt, Vt, B, H, d, S_part, pos0, S, Sp, eps=1e-6):
    _chk(qkv, BF16, "qkv"); _chk(cos, F32, "cos"); _chk(sin, F32, "sin")
    _need(d in (64, 128) and pos0 + S_part <= S and cos.shape[1] == d and cos.shape[0] >= pos0 + S_part, "qk_norm_rope_fwd: bad shape")
    D = H * d
    rows = _stream_rows(qkv, B, S, pos0, S_part).float()
    c, s = cos[pos0:pos0 + S_part], sin[pos0:pos0 + S_part]
    for j, (w, dst, dst_t) in enumerate(((wq, Q, Qt), (wk, K, Kt))):
        x = rows[..., j * D:(j + 1) * D].reshape(B, S_part, H, d)
        if w is not None:
            x = x * torch.rsqrt((x * x).mean(dim=-1, keepdim=True) + eps) * w.float()
        z = _rot(x, c, s)
        dst[:, :, pos0:pos0 + S_part] = z.permute(0, 2, 1, 3).to(BF16)
        if dst_t is not None:
            dst_t[:, :, :, pos0:pos0 + S_part] = z.permute(0, 2, 3, 1).to(BF16)
    v = rows[..., 2 * D:3 * D].reshape(B, S_part, H, d)
    Vt[:, :, :, pos0:pos0 + S_part] = v.permute(0, 2, 3, 1).to(BF16)


def qk_norm_rope_bwd_wgrad(dQ, dK, qkv, wq, wk, cos, sin, dqkv, B, H, d, S_part, pos0, S, gwq, gwk, accumulate=False, eps=1e-6):
    _chk(dQ, BF16, "dQ"); _chk(dK, BF16, "dK"); _chk(qkv, BF16, "qkv"); _chk(dqkv, BF16, "dqkv")
    _need((gwq is None or wq is not None) and (gwk is None or wk is not None), "qk_norm_rope_bwd_wgrad: a weight gradient needs its norm weight")
    _need(d in (64, 128) and pos0 + S_part <= S and S_part > 0, "qk_norm_rope_bwd: bad shape")
    D = H * d
    rows = _stream_rows(qkv, B, S, pos0, S_part).float()
    drows = _stream_rows(dqkv, B, S, pos0, S_part)
    c, s = cos[pos0:pos0 + S_part], sin[pos0:pos0 + S_part]
    for j, (w, g_heads, gw) in enumerate(((wq, dQ, gwq), (wk, dK, gwk))):
        x = rows[..., j * D:(j + 1) * D].reshape(B, S_part, H, d)
        g = g_heads[:, :, pos0:pos0 + S_part].float().permute(0, 2, 1, 3)
        dy = _rot_T(g, c, s)
        if w is not None:
            wf = w.float()
            r = torch.rsqrt((x * x).mean(dim=-1, keepdim=True) + eps)
            m = (x * wf * dy).mean(dim=-1, keepdim=True)
            dx = r * wf * dy - x * r ** 3 * m
            if gw is not None:
                _chk(gw, BF16, "gw")
                t = (dy * x * r).sum(dim=(0, 1, 2))
                gw.copy_((t + gw.float() if accumulate else t).to(BF16))
        else:
            dx = dy
        drows[..., j * D:(j + 1) * D] = dx.reshape(B, S_part, D).to(BF16)


def qk_norm_rope_bwd(dQ, dK, qkv, wq, wk, cos, sin, dqkv, B, H, d, S_part, pos0, S, eps=1e-6):
    qk_norm_rope_bwd_wgrad(dQ, dK, qkv, wq, wk, cos, sin, dqkv, B, H, d, S_part, pos0, S, None, None, eps=eps)


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _scores(q, k, B, S, scale, key_bias):
    sc = (q @ k.transpose(-1, -2)) * scale
    if key_bias is not None:
        sc = sc + key_bias[:, None, None, :S].float()
    return sc


def attn_fwd(Q, K, Vt, O, lse2, B, H, S, Sp, d, scale, key_bias=None, O_res=None):
    _chk(Q, BF16, "Q"); _chk(K, BF16, "K"); _chk(Vt, BF16, "Vt"); _chk(O, BF16, "O"); _chk(lse2, F32, "lse2")
    _rows(O, "O")
    _need(tuple(Q.shape) == (B, H, S, d) and tuple(K.shape) == (B, H, S, d) and tuple(Vt.shape) == (B, H, d, Sp) and Sp % 64 == 0 and Sp >= S, "attn_fwd: shapes")
    q, k, v = Q.float(), K.float(), Vt[..., :S].float().transpose(-1, -2)
    sc = _scores(q, k, B, S, scale, key_bias)
    lse2.copy_(torch.logsumexp(sc, dim=-1) * 1.4426950408889634)
    o = torch.softmax(sc, dim=-1) @ v                                      # [B, H, S, d]
    of = o.permute(0, 2, 1, 3).reshape(B, S, H * d)
    O.view(B, S, -1)[:, :, :H * d] = of.to(BF16)
    if O_res is not None:                  # the rounding residual of O (st355_attn_fwd_res); the emulated backward is exact autograd and has no use for it
        _chk(O_res, BF16, "O_res")
        _need(O_res.shape == O.shape and O_res.stride() == O.stride(), "attn_fwd: O_res must have O's layout")
        O_res.view(B, S, -1)[:, :, :H * d] = (of - of.to(BF16).float()).to(BF16)


def attn_bwd(Q, K, Qt, Kt, v_rows, O, dO, lse2, dQ, dK, dv_rows, B, H, S, Sp, d, scale, key_bias=None, O_res=None):
    for t, nm in ((Q, "Q"), (K, "K"), (v_rows, "v_rows"), (O, "O"), (dO, "dO"), (dQ, "dQ"), (dK, "dK"), (dv_rows, "dv_rows")):
        _chk(t, BF16, nm)
    for t, nm in ((v_rows, "v_rows"), (O, "O"), (dO, "dO"), (dv_rows, "dv_rows")):
        _rows(t, nm)
        _need(t.shape[0] == B * S and t.shape[1] >= H * d, f"attn_bwd: {nm} is {tuple(t.shape)}")
    heads = lambda t: t[:, :H * d].reshape(B, S, H, d).permute(0, 2, 1, 3).float()
    q, k, v = Q.float().requires_grad_(True), K.float().requires_grad_(True), heads(v_rows).requires_grad_(True)
    with torch.enable_grad():
        o = torch.softmax(_scores(q, k, B, S, scale, key_bias), dim=-1) @ v
        gq, gk, gv = torch.autograd.grad(o, (q, k, v), heads(dO))
    dQ.copy_(gq.to(BF16)); dK.copy_(gk.to(BF16))
    dv_rows[:, :H * d] = gv.permute(0, 2, 1, 3).reshape(B * S, H * d).to(BF16)


def attn_fwd_vrows(Q, K, v_rows, O, lse2, B, H, S, d, scale, key_bias=None):
    """attn_fwd with V row-major (token rows, head h at columns h*d): no V^T copy.  d = 128."""
    _chk(v_rows, BF16, "v_rows"); _rows(v_rows, "v_rows")
    _need(d == 128 and v_rows.shape[0] == B * S and v_rows.shape[1] >= H * d, "attn_fwd_vrows: shapes")
    Sp = (S + 63) // 64 * 64
    Vt = torch.zeros(B, H, d, Sp, dtype=BF16)
    Vt[..., :S] = v_rows[:, :H * d].reshape(B, S, H, d).permute(0, 2, 3, 1)
    attn_fwd(Q, K, Vt, O, lse2, B, H, S, Sp, d, scale, key_bias=key_bias)


def _qk_from_z(dz, z, rrms_part, w, cos_p, sin_p):
    """backward of RMSNorm(w) + pair rotation starting from the roped output z and the saved 1/rms (st355_qk_rope_norm_bwd): y = R^T z, x_hat = y / w,
    dy = R^T dz, dx = r (w dy - x_hat mean(dy y))"""
    dy = _rot_pairs_T(dz, cos_p, sin_p)
    if w is None:
        return dy
    y = _rot_pairs_T(z, cos_p, sin_p)
    wf = w.float()
    _need(float(wf.abs().min()) > 0.0, "qk_rope_norm_bwd: norm weights must be non-zero")
    xhat = y / wf
    return rrms_part[..., None] * (wf * dy - xhat * (dy * y).mean(dim=-1, keepdim=True))


def qk_rope_norm_bwd(dQ, dK, Q, K, rrms, wq, wk, cos, sin, dqkv, B, H, d, S_part, pos0, S):
    """from head-major dQ / dK: the dq / dk column blocks of dqkv for the S_part tokens at joint position pos0.  cos / sin here are the FULL-width [S, d] tables"""
    _need(d == 128, "qk_rope_norm_bwd: head_dim 128")
    D = H * d
    drows = _stream_rows(dqkv, B, S, pos0, S_part)
    rr = rrms.view(B, S, 2 * H)[:, pos0:pos0 + S_part]
    c, s = cos[pos0:pos0 + S_part, 0::2], sin[pos0:pos0 + S_part, 0::2]
    for j, (w, g, z) in enumerate(((wq, dQ, Q), (wk, dK, K))):
        gz = g[:, :, pos0:pos0 + S_part].float().permute(0, 2, 1, 3)
        zz = z[:, :, pos0:pos0 + S_part].float().permute(0, 2, 1, 3)
        dx = _qk_from_z(gz, zz, rr[..., j * H:(j + 1) * H], w, c, s)
        drows[..., j * D:(j + 1) * D] = dx.reshape(B, S_part, D).to(BF16)


def attn_bwd_rope(Q, K, v_rows, O, dO, lse2, rrms, wq_lo, wk_lo, wq_hi, wk_hi, split, cos_p, sin_p, dqkv, B, H, S, Sp, d, scale, key_bias=None):
    """attention backward with the RoPE + RMSNorm backward in the dQ / dK epilogues: dq, dk, dv land in the rows of dqkv [B*S, >= 3*H*d]; joint positions < split use
    the *_lo norm weights.  cos_p / sin_p: per-pair tables [S, 64]"""
    _need(d == 128 and 0 <= split <= S and dqkv.shape[1] >= 3 * H * d, "attn_bwd_rope: bad arguments")
    _need((wq_lo is None) == (wq_hi is None) and (wk_lo is None) == (wk_hi is None), "attn_bwd_rope: a norm weight needs both position ranges")
    _chk(rrms, F32, "rrms"); _chk(dqkv, BF16, "dqkv"); _chk(cos_p, F32, "cos_p"); _chk(sin_p, F32, "sin_p")
    D = H * d
    dQ = torch.empty(B, H, S, d, dtype=BF16); dK = torch.empty_like(dQ)
    attn_bwd(Q, K, None, None, v_rows, O, dO, lse2, dQ, dK, dqkv[:, 2 * D:3 * D], B, H, S, Sp, d, scale, key_bias=key_bias)
    rr = rrms.view(B, S, 2 * H)
    rows = dqkv.view(B, S, -1) if dqkv.is_contiguous() else _stream_rows(dqkv, B, S, 0, S)
    for lo, hi, wq, wk in ((0, split, wq_lo, wk_lo), (split, S, wq_hi, wk_hi)):
        if hi <= lo:
            continue
        c, s = cos_p[lo:hi], sin_p[lo:hi]
        for j, (w, g, z) in enumerate(((wq, dQ, Q), (wk, dK, K))):
            gz = g[:, :, lo:hi].float().permute(0, 2, 1, 3)
            zz = z[:, :, lo:hi].float().permute(0, 2, 1, 3)
            dx = _qk_from_z(gz, zz, rr[:, lo:hi, j * H:(j + 1) * H], w, c, s)
            rows[:, lo:hi, j * D:(j + 1) * D] = dx.reshape(B, hi - lo, D).to(BF16)


def attn_cross_fwd(Q, K, Vt, O, lse2, B, H, Sq, Sk, Skp, d, scale, key_bias=None, O_res=None):
    _chk(Q, BF16, "Q"); _chk(K, BF16, "K"); _chk(Vt, BF16, "Vt"); _chk(O, BF16, "O"); _chk(lse2, F32, "lse2")
    _rows(O, "O")
    _need(tuple(Q.shape) == (B, H, Sq, d) and tuple(K.shape) == (B, H, Sk, d) and tuple(Vt.shape) == (B, H, d, Skp) and Skp % 64 == 0 and Skp >= Sk, "attn_cross_fwd: shapes")
    sc = _scores(Q.float(), K.float(), B, Sk, scale, key_bias)
    lse2.copy_(torch.logsumexp(sc, dim=-1) * 1.4426950408889634)
    o = torch.softmax(sc, dim=-1) @ Vt[..., :Sk].float().transpose(-1, -2)
    of = o.permute(0, 2, 1, 3).reshape(B * Sq, H * d)
    O[:, :H * d] = of.to(BF16)
    if O_res is not None:
        _chk(O_res, BF16, "O_res")
        O_res[:, :H * d] = (of - of.to(BF16).float()).to(BF16)


def attn_cross_bwd(Q, K, Qt, Kt, v_rows, O, dO, lse2, dQ, dK, dv_rows, B, H, Sq, Sqp, Sk, Skp, d, scale, key_bias=None, O_res=None):
    for t, nm in ((Q, "Q"), (K, "K"), (v_rows, "v_rows"), (O, "O"), (dO, "dO"), (dQ, "dQ"), (dK, "dK"), (dv_rows, "dv_rows")):
        _chk(t, BF16, nm)
    for t, nm, n in ((v_rows, "v_rows", Sk), (O, "O", Sq), (dO, "dO", Sq), (dv_rows, "dv_rows", Sk)):
        _rows(t, nm)
        _need(t.shape[0] == B * n and t.shape[1] >= H * d, f"attn_cross_bwd: {nm} is {tuple(t.shape)}")
    heads = lambda t, n: t[:, :H * d].reshape(B, n, H, d).permute(0, 2, 1, 3).float()
    q, k, v = Q.float().requires_grad_(True), K.float().requires_grad_(True), heads(v_rows, Sk).requires_grad_(True)
    with torch.enable_grad():
        o = torch.softmax(_scores(q, k, B, Sk, scale, key_bias), dim=-1) @ v
        gq, gk, gv = torch.autograd.grad(o, (q, k, v), heads(dO, Sq))
    dQ.copy_(gq.to(BF16)); dK.copy_(gk.to(BF16))
    dv_rows[:, :H * d] = gv.permute(0, 2, 1, 3).reshape(B * Sk, H * d).to(BF16)


def head_split(src, B, H, d, S, want_x=True, want_xt=True, d_src=None):
    """src [B*S, H*d_src] (row stride free) -> (X [B,H,S,d] | None, Xt [B,H,d,Sp] | None, Sp); d_src < d: zero-padded heads"""
    _chk(src, BF16, "src"); _rows(src, "src")
    ds = d if d_src is None else d_src
    _need(src.shape[0] == B * S and src.shape[1] >= H * ds, f"head_split: src {tuple(src.shape)}")
    Sp = (S + 63) // 64 * 64
    x = torch.zeros(B, H, S, d, dtype=BF16)
    x[..., :ds] = src[:, :H * ds].reshape(B, S, H, ds).permute(0, 2, 1, 3)
    Xt = None
    if want_xt:
        Xt = torch.zeros(B, H, d, Sp, dtype=BF16)
        Xt[..., :S] = x.transpose(-1, -2)
    return (x if want_x else None), Xt, Sp


def head_merge(dX, dst, B, H, d, S, d_src=None):
    _chk(dX, BF16, "dX"); _chk(dst, BF16, "dst"); _rows(dst, "dst")
    ds = d if d_src is None else d_src
    _need(tuple(dX.shape) == (B, H, S, d) and dst.shape[0] == B * S and dst.shape[1] >= H * ds, "head_merge: shapes")
    dst[:, :H * ds] = dX[..., :ds].permute(0, 2, 1, 3).reshape(B * S, H * ds)
    return dst


def gelu_tanh(x):
    _chk(x, BF16, "x")
    return _gelu(x.float()).to(BF16)


# ------------------------------------------------------------------------------------------------
# UNet path: zero-bordered NHWC grid buffers ([B*(H+2)*(W+2) + 64, C]; st355.h), convolution-as-GEMM, GroupNorm, GEGLU, affine LayerNorm
# ------------------------------------------------------------------------------------------------
def grid_rows(B, H, W):
    return B * (H + 2) * (W + 2) + 64


def _interior(g, B, H, W, reads_borders=True):
    """grid [rows, C] -> fp32 [B, H, W, C] (the image positions).  reads_borders: the kernel being emulated reads border / tail rows as the zero padding of its
    shifted taps (convolutions, GroupNorm statistics over whole rows ...), so they must hold zero; gather-form kernels only touch image positions"""
    _chk(g, BF16, "grid"); _rows(g, "grid")
    _need(g.shape[0] == grid_rows(B, H, W), f"grid: {g.shape[0]} rows, a ({B},{H},{W}) grid has {grid_rows(B, H, W)}")
    n = B * (H + 2) * (W + 2)
    full = g[:n].reshape(B, H + 2, W + 2, g.shape[1]).float()
    if reads_borders:
        border = full.clone(); border[:, 1:H + 1, 1:W + 1] = 0
        _need(float(border.abs().max()) == 0.0 and float(g[n:].float().abs().max()) == 0.0, "grid: border / tail positions must hold zero")
    return full[:, 1:H + 1, 1:W + 1]


def _to_grid(img, out=None):
    """fp32 [B, H, W, C] -> grid buffer (borders and the 64 tail rows zero)"""
    B, H, W, Cn = img.shape
    g = torch.zeros(grid_rows(B, H, W), Cn, dtype=BF16) if out is None else out
    if out is not None:
        _chk(out, BF16, "out"); _need(tuple(out.shape) == (grid_rows(B, H, W), Cn), "grid out shape")
        out.zero_()
    g[:B * (H + 2) * (W + 2)].view(B, H + 2, W + 2, Cn)[:, 1:H + 1, 1:W + 1] = img.to(BF16)
    return g


def grid_zeros(B, H, W, C_, device, pool=False):
    return torch.zeros(grid_rows(B, H, W), C_, dtype=BF16)


def grid_from_nchw(x, Cpad):
    _chk(x, BF16, "x")
    B, Cn, H, W = x.shape
    img = torch.zeros(B, H, W, Cpad)
    img[..., :Cn] = x.float().permute(0, 2, 3, 1)
    return _to_grid(img)


def grid_to_nchw(g, B, Cn, H, W):
    return _interior(g, B, H, W)[..., :Cn].permute(0, 3, 1, 2).contiguous().to(BF16)


def tokens_to_grid(tokens, B, H, W, residual=None):
    _chk(tokens, BF16, "tokens"); _rows(tokens, "tokens")
    _need(tokens.shape[0] == B * H * W, "tokens_to_grid: rows")
    img = tokens.float().reshape(B, H, W, -1)
    if residual is not None:
        img = img + _interior(residual, B, H, W)
    return _to_grid(img)


def grid_to_tokens(g, B, H, W):
    return _interior(g, B, H, W).reshape(B * H * W, -1).to(BF16)


def conv(x, w, B, H, W, bias=None, img_add=None, residual=None, taps=9, out=None):
    _chk(x, BF16, "x"); _chk(w, BF16, "w")
    Cin, Cout = x.shape[1], w.shape[0]
    _need(taps in (1, 9) and w.shape[1] == taps * Cin and w.is_contiguous() and x.is_contiguous(), f"conv: weight {tuple(w.shape)} vs taps*Cin = {taps}*{Cin}")
    _need(Cin % 64 == 0 and Cout % 8 == 0, f"conv: Cin ({Cin}) must be a multiple of 64 and Cout ({Cout}) of 8")
    _al(x, 16, "x"); _al(w, 16, "w")
    xi = _interior(x, B, H, W)
    if taps == 9:
        wt = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        y = torch.nn.functional.conv2d(xi.permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1)
    else:
        y = xi @ w.float().t()
    if bias is not None:
        _chk(bias, BF16, "bias")
        y = y + bias.float()
    if img_add is not None:
        _chk(img_add, BF16, "img_add"); _rows(img_add, "img_add")
        _need(img_add.shape[0] >= B and img_add.shape[1] >= Cout, "conv: img_add rows")
        y = y + img_add[:B, :Cout].float()[:, None, None, :]
    if residual is not None:
        y = y + _interior(residual, B, H, W)
    return _to_grid(y, out)


def conv_wgrad(x, dy, dw, B, H, W, taps=9, accumulate=False):
    _chk(x, BF16, "x"); _chk(dy, BF16, "dy"); _chk(dw, BF16, "dw")
    Cin, Cout = x.shape[1], dy.shape[1]
    _need(tuple(dw.shape) == (Cout, taps * Cin) and dw.is_contiguous(), f"conv_wgrad: dw must be a contiguous [{Cout}, {taps * Cin}] tensor")
    xi, di = _interior(x, B, H, W), _interior(dy, B, H, W)
    if taps == 9:
        xp = torch.nn.functional.pad(xi, (0, 0, 1, 1, 1, 1))
        cols = [torch.einsum("bhwo,bhwi->oi", di, xp[:, ky:ky + H, kx:kx + W]) for ky in range(3) for kx in range(3)]
        val = torch.stack(cols, dim=1).reshape(Cout, 9 * Cin)
    else:
        val = torch.einsum("bhwo,bhwi->oi", di, xi)
    dw.copy_((val + dw.float() if accumulate else val).to(BF16))
    return dw


def _im2col(xi, stride, pad):
    """[B, H, W, C] -> [B, H/stride, W/stride, 9, C]: tap (ky, kx) of output (yo, xo) reads input (yo*stride + ky - pad, xo*stride + kx - pad), zero outside"""
    B, H, W, Cn = xi.shape
    Ho, Wo = H // stride, W // stride
    xp = torch.nn.functional.pad(xi, (0, 0, pad, 3, pad, 3))
    taps = [xp[:, ky:ky + Ho * stride:stride, kx:kx + Wo * stride:stride] for ky in range(3) for kx in range(3)]
    return torch.stack(taps, dim=3)


def im2col3x3(x, B, H, W, stride=1, pad=1):
    _need(stride in (1, 2) and pad in (0, 1), "im2col3x3: stride / pad")
    Cn = x.shape[1]
    Kpad = (9 * Cn + 63) // 64 * 64
    col = _im2col(_interior(x, B, H, W), stride, pad).reshape(B, H // stride, W // stride, 9 * Cn)
    return _to_grid(torch.nn.functional.pad(col, (0, Kpad - 9 * Cn)))


def col2im3x3(dcol, B, H, W, Cn, stride=1, pad=1):
    d = _interior(dcol, B, H // stride, W // stride, reads_borders=False)[..., :9 * Cn].reshape(B, H // stride, W // stride, 9, Cn)
    x = torch.zeros(B, H, W, Cn, requires_grad=True)
    with torch.enable_grad():
        (g,) = torch.autograd.grad(_im2col(x, stride, pad), x, d)
    return _to_grid(g)


def upsample2x(x, B, H, W):
    xi = _interior(x, B, H, W)
    return _to_grid(xi.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2))


def upsample2x_bwd(dy, B, H, W):
    d = _interior(dy, B, 2 * H, 2 * W)
    return _to_grid(d.reshape(B, H, 2, W, 2, -1).sum(dim=(2, 4)))


def _silu_grad(z):
    sg = torch.sigmoid(z)
    return sg * (1.0 + z * (1.0 - sg))


def groupnorm_fwd(x, gamma, beta, B, H, W, groups=32, eps=1e-5, silu=True, out_tokens=False):
    _chk(gamma, BF16, "gamma"); _chk(beta, BF16, "beta")
    xi = _interior(x, B, H, W)
    Cn = xi.shape[-1]
    _need(Cn % groups == 0, "groupnorm: C % groups")
    xg = xi.reshape(B, H * W, groups, Cn // groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 3), keepdim=True)
    rstd = torch.rsqrt(var + eps)
    y = ((xg - mean) * rstd).reshape(B, H, W, Cn) * gamma.float() + beta.float()
    if silu:
        y = y * torch.sigmoid(y)
    stats = torch.stack([mean.expand(B, 1, groups, Cn // groups).reshape(B, Cn), rstd.expand(B, 1, groups, Cn // groups).reshape(B, Cn)], dim=2).contiguous()
    return (y.reshape(B * H * W, Cn).to(BF16) if out_tokens else _to_grid(y)), stats


def groupnorm_bwd(dy, x, gamma, beta, stats, B, H, W, groups=32, silu=True, dy_tokens=False, dadd=None, dgamma=None, dbeta=None, accumulate_params=False):
    _chk(stats, F32, "stats")
    xi = _interior(x, B, H, W)
    Cn = xi.shape[-1]
    d = dy.float().reshape(B, H, W, Cn) if dy_tokens else _interior(dy, B, H, W)
    mean, rstd = stats[..., 0][:, None, None, :], stats[..., 1][:, None, None, :]
    xhat = (xi - mean) * rstd
    if silu:
        d = d * _silu_grad(xhat * gamma.float() + beta.float())
    for dst, val in ((dgamma, (d * xhat).sum(dim=(0, 1, 2))), (dbeta, d.sum(dim=(0, 1, 2)))):
        if dst is not None:
            _chk(dst, F32, "dgamma/dbeta")
            dst.copy_(val + dst if accumulate_params else val)
    g = (d * gamma.float()).reshape(B, H * W, groups, Cn // groups)
    xh = xhat.reshape(B, H * W, groups, Cn // groups)
    dx = rstd.reshape(B, 1, groups, Cn // groups) * (g - g.mean(dim=(1, 3), keepdim=True) - xh * (g * xh).mean(dim=(1, 3), keepdim=True))
    dx = dx.reshape(B, H, W, Cn)
    if dadd is not None:
        dx = dx + _interior(dadd, B, H, W)
    return _to_grid(dx)


def layernorm_fwd(x, weight, bias, eps=1e-5, out=None):
    _chk(x, BF16, "x"); _chk(weight, BF16, "weight"); _chk(bias, BF16, "bias"); _rows(x, "x")
    xhat, _ = _ln_stats(x.float(), eps)
    if out is None:
        out = torch.empty(x.shape, dtype=BF16)
    return _put(out, xhat * weight.float() + bias.float())


def layernorm_bwd(dy, x, weight, dres=None, eps=1e-5):
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(weight, BF16, "weight"); _rows(dy, "dy"); _rows(x, "x")
    xhat, rstd = _ln_stats(x.float(), eps)
    g = dy.float() * weight.float()
    dx = rstd * (g - g.mean(dim=1, keepdim=True) - xhat * (g * xhat).mean(dim=1, keepdim=True))
    if dres is not None:
        _chk(dres, BF16, "dres")
        dx = dx + dres.float()
    return dx.to(BF16)


def layernorm_param_grads(dy, x, dweight, dbias, eps=1e-5, accumulate=False):
    _chk(dy, BF16, "dy"); _chk(x, BF16, "x"); _chk(dweight, F32, "dweight"); _chk(dbias, F32, "dbias")
    xhat, _ = _ln_stats(x.float(), eps)
    for dst, val in ((dweight, (dy.float() * xhat).sum(0)), (dbias, dy.float().sum(0))):
        dst.copy_(val + dst if accumulate else val)


def _gelu_erf(g):
    return 0.5 * g * (1.0 + torch.erf(g * 0.7071067811865476))


def geglu_fwd(h):
    _chk(h, BF16, "h"); _rows(h, "h")
    F_ = h.shape[1] // 2
    return (h[:, :F_].float() * _gelu_erf(h[:, F_:].float())).to(BF16)


def geglu_bwd(h, dout):
    _chk(h, BF16, "h"); _chk(dout, BF16, "dout")
    F_ = h.shape[1] // 2
    v, g, d = h[:, :F_].float(), h[:, F_:].float(), dout.float()
    gg = 0.5 * (1.0 + torch.erf(g * 0.7071067811865476)) + g * 0.3989422804014327 * torch.exp(-0.5 * g * g)
    return torch.cat([d * _gelu_erf(g), d * v * gg], dim=1).to(BF16)


def softmax_rows_(x, scale=1.0):
    _chk(x, BF16, "x"); _rows(x, "x")
    x.copy_(torch.softmax(x.float() * scale, dim=1).to(BF16))
    return x


def softmax_rows_bwd_(p, dp, scale=1.0):
    _chk(p, BF16, "p"); _chk(dp, BF16, "dp")
    _need(_rows(p, "p") == _rows(dp, "dp"), "softmax_rows_bwd: p and dp must share a row stride")
    pf, df = p.float(), dp.float()
    dp.copy_((scale * pf * (df - (df * pf).sum(dim=1, keepdim=True))).to(BF16))
    return dp


BLOCK_CALLS = {}          # how often each emulated block-level entry point ran (tests assert the engines really take them)


def block_pixart_fwd(**a):
    """st355_block_pixart_fwd (csrc/blocks.hip) restated over the emulated entry points, in its order, writing the caller's buffers: what the C function does with
    the struct fields.  It checks the HOST side of the boundary on a CPU box (which buffer goes into which field, shapes, the zero-pad contract of Vt / V2t); the
    C++ sequencing itself is checked on the MI355X against the host-side sequencing (tests/test_pixart_model_gpu.py)."""
    A = SimpleNamespace(**a)
    BLOCK_CALLS["pixart_fwd"] = BLOCK_CALLS.get("pixart_fwd", 0) + 1
    B, S, Sk, H, D, hd = A.B, A.S, A.Sk, A.H, A.D, A.d_pad
    Dp = H * hd
    Sp, Skp = (S + 63) // 64 * 64, (Sk + 63) // 64 * 64
    assert A.mod.shape == (B, 6 * D) and A.mod_stride == A.mod.stride(0)
    for t, sh in ((A.n1, (B * S, D)), (A.qkv, (B * S, 3 * Dp)), (A.Q, (B, H, S, hd)), (A.K, (B, H, S, hd)), (A.O, (B * S, Dp)), (A.lse, (B, H, S)), (A.h1, (B * S, D)),
                  (A.q2, (B * S, Dp)), (A.kv, (B * Sk, 2 * Dp)), (A.Q2, (B, H, S, hd)), (A.K2, (B, H, Sk, hd)), (A.O2, (B * S, Dp)), (A.lse_x, (B, H, S)),
                  (A.h2, (B * S, D)), (A.n2, (B * S, D)), (A.act, (B * S, 4 * D)), (A.Vt, (B, H, hd, Sp)), (A.V2t, (B, H, hd, Skp)), (A.out, (B * S, D))):
        assert tuple(t.shape) == sh and t.is_contiguous(), (tuple(t.shape), sh)
    assert Sp == S or float(A.Vt[..., S:].abs().max()) == 0, "Vt: the columns beyond S must be zero on entry"
    assert Skp == Sk or float(A.V2t[..., Sk:].abs().max()) == 0, "V2t: the columns beyond Sk must be zero on entry"
    m = [A.mod[:, k * D:(k + 1) * D] for k in range(6)]
    ln_modulate_fwd(A.h, m[1], m[0], S, out=A.n1)
    gemm(A.n1, A.w_qkv, bias=A.b_qkv, out=A.qkv)
    A.Q.copy_(head_split(A.qkv[:, :Dp], B, H, hd, S, want_xt=False)[0])
    A.K.copy_(head_split(A.qkv[:, Dp:2 * Dp], B, H, hd, S, want_xt=False)[0])
    A.Vt.copy_(head_split(A.qkv[:, 2 * Dp:], B, H, hd, S, want_x=False)[1])
    attn_fwd(A.Q, A.K, A.Vt, A.O, A.lse, B, H, S, Sp, hd, A.scale)
    gemm(A.O, A.w_out1, bias=A.b_out1, epilogue=EPI_GATE_RESIDUAL, gate=m[2], aux_in=A.h, rows_per_batch=S, aux_out=A.ya, out=A.h1)
    gemm(A.h1, A.w_q2, bias=A.b_q2, out=A.q2)
    gemm(A.ctx, A.w_kv2, bias=A.b_kv2, out=A.kv)
    A.Q2.copy_(head_split(A.q2, B, H, hd, S, want_xt=False)[0])
    A.K2.copy_(head_split(A.kv[:, :Dp], B, H, hd, Sk, want_xt=False)[0])
    A.V2t.copy_(head_split(A.kv[:, Dp:], B, H, hd, Sk, want_x=False)[1])
    attn_cross_fwd(A.Q2, A.K2, A.V2t, A.O2, A.lse_x, B, H, S, Sk, Skp, hd, A.scale, key_bias=A.key_bias)
    gemm(A.O2, A.w_out2, bias=A.b_out2, epilogue=EPI_ADD, aux_in=A.h1, out=A.h2)
    ln_modulate_fwd(A.h2, m[4], m[3], S, out=A.n2)
    gemm(A.n2, A.w_ff1, bias=A.b_ff1, epilogue=EPI_GELU, aux_out=A.pre, out=A.act)
    gemm(A.act, A.w_ff2, bias=A.b_ff2, epilogue=EPI_GATE_RESIDUAL, gate=m[5], aux_in=A.h2, rows_per_batch=S, aux_out=A.yf, out=A.out)


def block_pixart_bwd(**a):
    """st355_block_pixart_bwd restated over the emulated entry points (see block_pixart_fwd)"""
    A = SimpleNamespace(**a)
    BLOCK_CALLS["pixart_bwd"] = BLOCK_CALLS.get("pixart_bwd", 0) + 1
    B, S, Sk, H, D, hd = A.B, A.S, A.Sk, A.H, A.D, A.d_pad
    Dp = H * hd
    Sp, Skp = (S + 63) // 64 * 64, (Sk + 63) // 64 * 64
    for t, sh in ((A.dyf, (B * S, D)), (A.dpre, (B * S, 4 * D)), (A.dn2, (B * S, D)), (A.d2, (B * S, D)), (A.dO2, (B * S, Dp)), (A.dq2, (B * S, Dp)),
                  (A.dkv, (B * Sk, 2 * Dp)), (A.d1, (B * S, D)), (A.dya, (B * S, D)), (A.dO, (B * S, Dp)), (A.dqkv, (B * S, 3 * Dp)), (A.dn1, (B * S, D)),
                  (A.d_in, (B * S, D)), (A.d_out, (B * S, D))):
        assert tuple(t.shape) == sh and t.is_contiguous(), (tuple(t.shape), sh)
    assert A.dQ.numel() >= B * H * S * hd and A.dK.numel() >= B * H * max(S, Sk) * hd
    m = [A.mod[:, k * D:(k + 1) * D] for k in range(6)]
    scale_cols(A.d_out, m[5], S, out=A.dyf)
    gemm(A.dyf, A.wT_ff2, epilogue=EPI_MUL_GELU_GRAD, aux_in=A.pre, out=A.dpre)
    gemm(A.dpre, A.wT_ff1, out=A.dn2)
    ln_modulate_bwd(A.dn2, A.h2, m[4], S, dres=A.d_out, out=A.d2)
    gemm(A.d2, A.wT_out2, out=A.dO2)
    dQ = A.dQ.view(-1)[:B * H * S * hd].view(B, H, S, hd); dKx = A.dK.view(-1)[:B * H * Sk * hd].view(B, H, Sk, hd)
    attn_cross_bwd(A.Q2, A.K2, None, None, A.kv[:, Dp:], A.O2, A.dO2, A.lse_x, dQ, dKx, A.dkv[:, Dp:], B, H, S, Sp, Sk, Skp, hd, A.scale, key_bias=A.key_bias)
    head_merge(dQ, A.dq2, B, H, hd, S)
    head_merge(dKx, A.dkv[:, :Dp], B, H, hd, Sk)
    gemm(A.dq2, A.wT_q2, epilogue=EPI_ADD, aux_in=A.d2, out=A.d1)
    scale_cols(A.d1, m[2], S, out=A.dya)
    gemm(A.dya, A.wT_out1, out=A.dO)
    dKs = A.dK.view(-1)[:B * H * S * hd].view(B, H, S, hd)
    attn_bwd(A.Q, A.K, None, None, A.qkv[:, 2 * Dp:], A.O, A.dO, A.lse, dQ, dKs, A.dqkv[:, 2 * Dp:], B, H, S, Sp, hd, A.scale)
    head_merge(dQ, A.dqkv[:, :Dp], B, H, hd, S)
    head_merge(dKs, A.dqkv[:, Dp:2 * Dp], B, H, hd, S)
    gemm(A.dqkv, A.wT_qkv, out=A.dn1)
    ln_modulate_bwd(A.dn1, A.h, m[1], S, dres=A.d1, out=A.d_in)


def _sample_rows(buf, b, lo, rows, S):
    """rows [lo, lo + rows) of sample b of a joint [B * S, C] buffer (a contiguous 2-D slice)"""
    return buf[b * S + lo:b * S + lo + rows]


def block_sd3_joint_fwd(**a):
    """st355_block_sd3_joint_fwd (csrc/blocks.hip) restated over the emulated entry points, writing the caller's buffers.  Joint-buffer operands are walked one
    sample at a time here (the C function's single / per-sample / compact-copy problem forms are launch shapes of the same arithmetic).  See block_pixart_fwd."""
    A = SimpleNamespace(**a)
    BLOCK_CALLS["sd3_fwd"] = BLOCK_CALLS.get("sd3_fwd", 0) + 1
    B, Si, St, H, D, hd = A.B, A.Si, A.St, A.H, A.D, A.hd
    S = Si + St
    Sp = (S + 63) // 64 * 64
    last = bool(A.last)
    assert D == H * hd and A.mod_stride == A.mod_img.stride(0) == A.mod_txt.stride(0)
    for t, sh in ((A.n_img, (B * Si, D)), (A.n_txt, (B * St, D)), (A.qkv, (B * S, 3 * D)), (A.Q, (B, H, S, hd)), (A.K, (B, H, S, hd)), (A.O, (B * S, D)),
                  (A.lse2, (B, H, S)), (A.x1_img, (B * Si, D)), (A.hpre_img, (B * Si, 4 * D)), (A.n2_img, (B * Si, D)), (A.h_img, (B * Si, 4 * D)),
                  (A.Vt, (B, H, hd, Sp)), (A.out_img, (B * Si, D))):
        assert tuple(t.shape) == sh and t.is_contiguous(), (tuple(t.shape), sh)
    assert Sp == S or float(A.Vt[..., S:].abs().max()) == 0, "Vt: the columns beyond S must be zero on entry"
    for rows, c in ((Si, A.c_img), (St, A.c_txt)):
        assert not (B > 1 and rows % 256) or (c is not None and c.numel() >= B * rows * 3 * D), "a stream that is not tile-aligned needs its compact-copy scratch"
    mi = A.mod_img[:, :6 * D]
    mt = A.mod_txt[:, :(2 if last else 6) * D]
    ln_modulate_fwd(A.img, mi[:, D:2 * D], mi[:, :D], Si, out=A.n_img)
    if last:
        ln_modulate_fwd(A.txt, mt[:, :D], mt[:, D:2 * D], St, out=A.n_txt)
    else:
        ln_modulate_fwd(A.txt, mt[:, D:2 * D], mt[:, :D], St, out=A.n_txt)
    if A.K2_qkv:
        gemm(A.n_img, A.A_qkv, out=A.T_img)
    if A.K2_aqkv:
        gemm(A.n_txt, A.A_aqkv, out=A.T_txt)
    for b in range(B):
        kw = dict(a2=A.T_img[b * Si:(b + 1) * Si], b2=A.Bb_qkv) if A.K2_qkv else {}
        gemm(A.n_img[b * Si:(b + 1) * Si], A.w_qkv, bias=A.b_qkv, out=_sample_rows(A.qkv, b, 0, Si, S), **kw)
        kw = dict(a2=A.T_txt[b * St:(b + 1) * St], b2=A.Bb_aqkv) if A.K2_aqkv else {}
        gemm(A.n_txt[b * St:(b + 1) * St], A.w_add_qkv, bias=A.b_add_qkv, out=_sample_rows(A.qkv, b, Si, St, S), **kw)
    qk_norm_rope_fwd(A.qkv, A.norm_q, A.norm_k, A.cos, A.sin, A.Q, A.K, None, None, A.Vt, B, H, hd, Si, 0, S, Sp)
    qk_norm_rope_fwd(A.qkv, A.norm_added_q, A.norm_added_k, A.cos, A.sin, A.Q, A.K, None, None, A.Vt, B, H, hd, St, Si, S, Sp)
    attn_fwd(A.Q, A.K, A.Vt, A.O, A.lse2, B, H, S, Sp, hd, A.scale)
    for b in range(B):
        O_i, O_t = _sample_rows(A.O, b, 0, Si, S), _sample_rows(A.O, b, Si, St, S)
        si, st = slice(b * Si, (b + 1) * Si), slice(b * St, (b + 1) * St)
        kw = {}
        if A.K2_out:
            gemm(O_i, A.A_out, out=A.T_o[si])
            kw = dict(a2=A.T_o[si], b2=A.Bb_out)
        gemm(O_i, A.w_out, bias=A.b_out, epilogue=EPI_GATE_RESIDUAL, aux_in=A.img[si], gate=mi[b:b + 1, 2 * D:3 * D], rows_per_batch=Si,
             aux_out=None if A.ya_img is None else A.ya_img[si], out=A.x1_img[si], **kw)
        if not last:
            kw = {}
            if A.K2_aout:
                gemm(O_t, A.A_aout, out=A.T_ao[st])
                kw = dict(a2=A.T_ao[st], b2=A.Bb_aout)
            gemm(O_t, A.w_add_out, bias=A.b_add_out, epilogue=EPI_GATE_RESIDUAL, aux_in=A.txt[st], gate=mt[b:b + 1, 2 * D:3 * D], rows_per_batch=St,
                 aux_out=None if A.ya_txt is None else A.ya_txt[st], out=A.x1_txt[st], **kw)
    ln_modulate_fwd(A.x1_img, mi[:, 4 * D:5 * D], mi[:, 3 * D:4 * D], Si, out=A.n2_img)
    gemm(A.n2_img, A.w_ff1, bias=A.b_ff1, epilogue=EPI_GELU, aux_out=A.hpre_img, out=A.h_img)
    gemm(A.h_img, A.w_ff2, bias=A.b_ff2, epilogue=EPI_GATE_RESIDUAL, aux_in=A.x1_img, gate=mi[:, 5 * D:6 * D], rows_per_batch=Si, aux_out=A.yf_img, out=A.out_img)
    if not last:
        ln_modulate_fwd(A.x1_txt, mt[:, 4 * D:5 * D], mt[:, 3 * D:4 * D], St, out=A.n2_txt)
        gemm(A.n2_txt, A.w_ffc1, bias=A.b_ffc1, epilogue=EPI_GELU, aux_out=A.hpre_txt, out=A.h_txt)
        gemm(A.h_txt, A.w_ffc2, bias=A.b_ffc2, epilogue=EPI_GATE_RESIDUAL, aux_in=A.x1_txt, gate=mt[:, 5 * D:6 * D], rows_per_batch=St, aux_out=A.yf_txt, out=A.out_txt)


def block_sd3_joint_bwd(**a):
    """st355_block_sd3_joint_bwd restated over the emulated entry points (see block_sd3_joint_fwd)"""
    A = SimpleNamespace(**a)
    BLOCK_CALLS["sd3_bwd"] = BLOCK_CALLS.get("sd3_bwd", 0) + 1
    B, Si, St, H, D, hd = A.B, A.Si, A.St, A.H, A.D, A.hd
    S = Si + St
    Sp = (S + 63) // 64 * 64
    last = bool(A.last)
    for t, sh in ((A.g_img, (B * Si, D)), (A.dh_img, (B * Si, 4 * D)), (A.dn2_img, (B * Si, D)), (A.dx1_img, (B * Si, D)), (A.dx1g_img, (B * Si, D)),
                  (A.dO, (B * S, D)), (A.dqkv, (B * S, 3 * D)), (A.dQ, (B, H, S, hd)), (A.dK, (B, H, S, hd)), (A.d_img, (B * Si, D))):
        assert tuple(t.shape) == sh and t.is_contiguous(), (tuple(t.shape), sh)
    for rows, c in ((Si, A.c_img), (St, A.c_txt)):
        assert not (B > 1 and rows % 256) or (c is not None and c.numel() >= B * rows * 3 * D), "a stream that is not tile-aligned needs its compact-copy scratch"
    mi = A.mod_img[:, :6 * D]
    mt = A.mod_txt[:, :(2 if last else 6) * D]
    scale_cols(A.d_img, mi[:, 5 * D:6 * D], Si, out=A.g_img)
    gemm(A.g_img, A.wT_ff2, epilogue=EPI_MUL_GELU_GRAD, aux_in=A.hpre_img, out=A.dh_img)
    gemm(A.dh_img, A.wT_ff1, out=A.dn2_img)
    if not last:
        scale_cols(A.d_txt, mt[:, 5 * D:6 * D], St, out=A.g_txt)
        gemm(A.g_txt, A.wT_ffc2, epilogue=EPI_MUL_GELU_GRAD, aux_in=A.hpre_txt, out=A.dh_txt)
        gemm(A.dh_txt, A.wT_ffc1, out=A.dn2_txt)
        _, dxg = ln_modulate_bwd(A.dn2_txt, A.x1_txt, mt[:, 4 * D:5 * D], St, dres=A.d_txt, gate=mt[:, 2 * D:3 * D], want_gated=True, out=A.dx1_txt)
        A.dx1g_txt.copy_(dxg)
    _, dxg = ln_modulate_bwd(A.dn2_img, A.x1_img, mi[:, 4 * D:5 * D], Si, dres=A.d_img, gate=mi[:, 2 * D:3 * D], want_gated=True, out=A.dx1_img)
    A.dx1g_img.copy_(dxg)
    if last:
        assert float(A.dO.view(B, S, D)[:, Si:].abs().max()) == 0, "dO: a context_pre_only block needs the text rows zero-filled on entry"
    if A.K2_out:
        gemm(A.dx1g_img, A.Bbt_out, out=A.U_o)
    if not last and A.K2_aout:
        gemm(A.dx1g_txt, A.Bbt_aout, out=A.U_ao)
    for b in range(B):
        si, st = slice(b * Si, (b + 1) * Si), slice(b * St, (b + 1) * St)
        kw = dict(a2=A.U_o[si], b2=A.At_out) if A.K2_out else {}
        gemm(A.dx1g_img[si], A.wT_out, out=_sample_rows(A.dO, b, 0, Si, S), **kw)
        if not last:
            kw = dict(a2=A.U_ao[st], b2=A.At_aout) if A.K2_aout else {}
            gemm(A.dx1g_txt[st], A.wT_add_out, out=_sample_rows(A.dO, b, Si, St, S), **kw)
    attn_bwd(A.Q, A.K, None, None, A.qkv[:, 2 * D:], A.O, A.dO, A.lse2, A.dQ, A.dK, A.dqkv[:, 2 * D:], B, H, S, Sp, hd, A.scale)
    qk_norm_rope_bwd(A.dQ, A.dK, A.qkv, A.norm_q, A.norm_k, A.cos, A.sin, A.dqkv, B, H, hd, Si, 0, S)
    qk_norm_rope_bwd(A.dQ, A.dK, A.qkv, A.norm_added_q, A.norm_added_k, A.cos, A.sin, A.dqkv, B, H, hd, St, Si, S)
    for rows, lo, c in ((Si, 0, A.c_img), (St, Si, A.c_txt)):
        if B > 1 and rows % 256:                      # the compact copy the caller reads the stream's rows of dqkv from afterwards
            c.view(-1)[:B * rows * 3 * D].view(B, rows, 3 * D).copy_(A.dqkv.view(B, S, 3 * D)[:, lo:lo + rows])
    for b in range(B):
        si, st = slice(b * Si, (b + 1) * Si), slice(b * St, (b + 1) * St)
        dq_i, dq_t = _sample_rows(A.dqkv, b, 0, Si, S), _sample_rows(A.dqkv, b, Si, St, S)
        if A.K2_qkv:
            gemm(dq_i, A.Bbt_qkv, out=A.U_qkv[si])
        if A.K2_aqkv:
            gemm(dq_t, A.Bbt_aqkv, out=A.U_aqkv[st])
        if A.need_input_grads:
            kw = dict(a2=A.U_qkv[si], b2=A.At_qkv) if A.K2_qkv else {}
            gemm(dq_i, A.wT_qkv, out=A.dn_img[si], **kw)
            kw = dict(a2=A.U_aqkv[st], b2=A.At_aqkv) if A.K2_aqkv else {}
            gemm(dq_t, A.wT_add_qkv, out=A.dn_txt[st], **kw)
    if A.need_input_grads:
        ln_modulate_bwd(A.dn_img, A.img, mi[:, D:2 * D], Si, dres=A.dx1_img, out=A.d_img_out)
        ln_modulate_bwd(A.dn_txt, A.txt, mt[:, :D] if last else mt[:, D:2 * D], St, dres=None if last else A.dx1_txt, out=A.d_txt_out)
    if getattr(A, "dmod_img", None) is not None:
        # the fused-statistics form (csrc/stats.hip): fp32 sums over the bf16 tensors the entry wrote; LN(x) in fp32 (the kernel never rounds it)
        assert A.need_input_grads and A.dmod_stride == A.dmod_img.stride(0) == A.dmod_txt.stride(0)
        xhat = lambda x: torch.nn.functional.layer_norm(x.float(), (D,), eps=1e-6)
        per_b = lambda t, rows: t.view(B, rows, -1).sum(dim=1)

        def stream(dm, rows, d_in, yf, g, dh, dn2, x1, dx1, ya, dx1g, gb2, gb1, gbo, lo, gbq, dn, x_in, k_shift, k_scale, mlp):
            if mlp:
                dm[:, 5 * D:6 * D] = per_b(d_in.float() * yf.float(), rows)
                gb2.copy_(g.float().sum(dim=0)); gb1.copy_(dh.float().sum(dim=0))
                dm[:, 3 * D:4 * D] = per_b(dn2.float(), rows)
                dm[:, 4 * D:5 * D] = per_b(dn2.float() * xhat(x1), rows)
                dm[:, 2 * D:3 * D] = per_b(dx1.float() * ya.float(), rows)
                gbo.copy_(dx1g.float().sum(dim=0))
            gbq.copy_(A.dqkv.view(B, S, 3 * D)[:, lo:lo + rows].float().sum(dim=(0, 1)))
            dm[:, k_shift * D:(k_shift + 1) * D] = per_b(dn.float(), rows)
            dm[:, k_scale * D:(k_scale + 1) * D] = per_b(dn.float() * xhat(x_in), rows)

        stream(A.dmod_img, Si, A.d_img, A.yf_img, A.g_img, A.dh_img, A.dn2_img, A.x1_img, A.dx1_img, A.ya_img, A.dx1g_img, A.gb_ff2, A.gb_ff1, A.gb_out, 0, A.gb_qkv,
               A.dn_img, A.img, 0, 1, True)
        stream(A.dmod_txt, St, A.d_txt, A.yf_txt, A.g_txt, A.dh_txt, A.dn2_txt, A.x1_txt, A.dx1_txt, A.ya_txt, A.dx1g_txt, A.gb_ffc2, A.gb_ffc1, A.gb_add_out, Si,
               A.gb_add_qkv, A.dn_txt, A.txt, 1 if last else 0, 0 if last else 1, not last)


_EMULATED = ("qk_rope", "attn_fwd_vrows", "attn_bwd_rope", "qk_rope_norm_bwd", "grid_rows", "grid_zeros", "grid_from_nchw", "grid_to_nchw", "tokens_to_grid", "grid_to_tokens", "conv", "conv_wgrad", "im2col3x3", "col2im3x3", "upsample2x",
             "upsample2x_bwd", "groupnorm_fwd", "groupnorm_bwd", "layernorm_fwd", "layernorm_bwd", "layernorm_param_grads", "geglu_fwd", "geglu_bwd", "softmax_rows_",
             "softmax_rows_bwd_", "attn_cross_fwd", "attn_cross_bwd", "head_split", "head_merge", "gelu_tanh", "gemm", "gemm_grouped", "gemm_tn", "colsum_prod", "transpose", "skinny_tn", "skinny_tn_multi", "lora_pack", "flow_noise_mix", "ddpm_noise_mix", "flux_pack", "flux_unpack", "mse_loss", "cond_loss", "adamw_ema_step", "adamw_bf16_sr_step", "ema_update", "grad_norm", "grad_clamp_", "grad_clip_norm_", "timestep_proj", "patchify", "unpatchify", "silu", "silu_bwd",
             "add", "scale_cols", "scale_cols_stats", "colsum_rows", "ln_modulate_bwd_stats", "gather_rows", "scatter_rows", "ln_modulate_fwd", "ln_modulate_bwd", "layer_norm_xhat", "qk_norm_rope_fwd", "qk_norm_rope_bwd",
             "qk_norm_rope_bwd_wgrad", "attn_fwd", "attn_bwd", "block_pixart_fwd", "block_pixart_bwd", "block_sd3_joint_fwd", "block_sd3_joint_bwd")


def install(monkeypatch):
    """replace the emulated wrappers of simpletuner_amd.ops (pytest's monkeypatch restores them); every other wrapper still raises on host tensors"""
    from simpletuner_amd import ops
    g = globals()
    for name in _EMULATED:
        monkeypatch.setattr(ops, name, g[name])
    return ops
