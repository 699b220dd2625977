"""GPU parity of the round-6 kernels, through the C ABI: the fused token-axis statistics (csrc/stats.hip), the segmented-contraction weight-gradient GEMM
(st355_gemm_tn_seg_bf16), the ragged key tail behind the 64-row attention dQ kernel, the EMA update inside the AdamW launch.  References: the unfused entry points
(bit-exact where the arithmetic is the same) and plain fp32 torch on the same seeded inputs (tolerances stated per assert)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def dev():
    return torch.device("cuda:0")


def rel(a, ref):
    a = a.double(); ref = ref.double()
    return ((a - ref).norm() / (ref.norm() + 1e-30)).item()


@pytest.fixture(scope="module")
def ops():
    from simpletuner_amd import ops as o

    return o


@pytest.mark.parametrize("B,S,D", [(3, 231, 1536), (2, 4096, 1536), (2, 300, 3072), (4, 64, 256), (1, 1000, 1152)])
def test_ln_modulate_bwd_with_fused_modulation_gate_and_bias_sums(ops, B, S, D):
    """st355_ln_modulate_bwd_stats: dx / dxg = st355_ln_modulate_bwd's up to FMA contraction (see below); d shift = sum dy, d scale = sum dy * LN(x) (LN in fp32, never rounded),
    d gate = sum dx * y (dx as stored), d bias = sum dxg (as stored) against fp32 torch sums of the same bf16 tensors: fp32 summation order only (<= 2e-5 rel)."""
    torch.manual_seed(61)
    d_ = dev()
    x = torch.randn(B * S, D, device=d_).to(BF16); dy = (torch.randn(B * S, D, device=d_) * 0.3).to(BF16)
    dres = torch.randn(B * S, D, device=d_).to(BF16); ya = torch.randn(B * S, D, device=d_).to(BF16)
    mod = (torch.randn(B, 3 * D, device=d_) * 0.3).to(BF16)
    scale, gate = mod[:, :D], mod[:, 2 * D:]
    dx0, dxg0 = ops.ln_modulate_bwd(dy, x, scale, S, dres=dres, gate=gate, want_gated=True)
    dmod = torch.full((B, 4 * D), 7.0, device=d_, dtype=F32)                       # destinations are OVERWRITTEN (strided slices of one buffer)
    gb = torch.full((D,), 3.0, device=d_, dtype=BF16)
    dx, dxg = ops.ln_modulate_bwd_stats(dy, x, scale, S, dmod[:, :D], dmod[:, D:2 * D], dres=dres, gate=gate, y_branch=ya, d_gate=dmod[:, 2 * D:3 * D], d_bias=gb,
                                        want_gated=True)
    # the same formula as k_ln_mod_bwd; hipcc contracts the two kernels' multiply-adds differently (aggressive FMA fusion), so a few outputs land on the
    # neighbouring bf16 value: rel-L2 <= 1e-3 (stated; measured ~1e-4), never more than one bf16 ulp apart
    for a, b in ((dx, dx0), (dxg, dxg0)):
        assert rel(a, b) < 1e-3
        assert float(((a.float() - b.float()).abs() / b.float().abs().clamp_min(1e-3)).max()) < 2 ** -6
    xh = torch.nn.functional.layer_norm(x.float(), (D,), eps=1e-6)
    pb = lambda t: t.view(B, S, D).sum(1)
    assert rel(dmod[:, :D], pb(dy.float())) < 2e-5
    assert rel(dmod[:, D:2 * D], pb(dy.float() * xh)) < 2e-5
    assert rel(dmod[:, 2 * D:3 * D], pb(dx.float() * ya.float())) < 2e-5
    assert float(dmod[:, 3 * D:].min()) == 7.0 and float(dmod[:, 3 * D:].max()) == 7.0          # nothing written next to the slices
    ref_b = dxg.float().sum(0)
    assert rel(gb.float(), ref_b) < 4e-3                                              # one rounding of the sum to bf16
    # the two-sum form (no gate statistics), no residual, fp32 bias row
    d2 = torch.zeros(B, 2 * D, device=d_, dtype=F32)
    dx1, _ = ops.ln_modulate_bwd_stats(dy, x, scale, S, d2[:, D:], d2[:, :D])
    dx1_ref, _ = ops.ln_modulate_bwd(dy, x, scale, S)
    assert rel(dx1, dx1_ref) < 1e-3
    assert rel(d2[:, D:], pb(dy.float())) < 2e-5 and rel(d2[:, :D], pb(dy.float() * xh)) < 2e-5


@pytest.mark.parametrize("B,S,N", [(3, 231, 1536), (2, 4096, 1536), (8, 154, 3072), (2, 100, 520)])
def test_scale_cols_with_fused_gate_and_bias_sums_and_strided_colsum_rows(ops, B, S, N):
    torch.manual_seed(62)
    d_ = dev()
    x = torch.randn(B * S, N, device=d_).to(BF16); y = torch.randn(B * S, N, device=d_).to(BF16)
    gate = (torch.randn(B, 2 * N, device=d_) * 0.5).to(BF16)[:, N:]
    g0 = ops.scale_cols(x, gate, S)
    dg = torch.zeros(B, N, device=d_, dtype=F32); gb = torch.zeros(N, device=d_, dtype=BF16)
    g = ops.scale_cols_stats(x, gate, S, y_branch=y, d_gate=dg, d_bias=gb)
    assert torch.equal(g, g0)
    assert rel(dg, (x.float() * y.float()).view(B, S, N).sum(1)) < 2e-5
    assert rel(gb.float(), g0.float().sum(0)) < 4e-3
    # plain column sums of a stream's rows of a joint [B, S + T, N] buffer, in place: one bf16 row over all samples, fp32 per-sample rows, accumulate
    T = 37
    joint = torch.randn(B * (S + T), N, device=d_).to(BF16)
    rows = joint.view(B, S + T, N)[:, T:]
    one = torch.zeros(N, device=d_, dtype=F32)
    ops.colsum_rows(joint[T:], S, S + T, B, one)
    assert rel(one, rows.float().sum((0, 1))) < 2e-5
    ops.colsum_rows(joint[T:], S, S + T, B, one, accumulate=True)
    assert rel(one, 2 * rows.float().sum((0, 1))) < 2e-5
    per = torch.zeros(B, N, device=d_, dtype=F32)
    ops.colsum_rows(joint[T:], S, S + T, B, per, per_batch=True)
    assert rel(per, rows.float().sum(1)) < 2e-5
    b16 = torch.zeros(N, device=d_, dtype=BF16)
    ops.colsum_rows(joint[T:], S, S + T, B, b16)
    assert rel(b16.float(), rows.float().sum((0, 1))) < 4e-3


@pytest.mark.parametrize("B,rows,lo,S,P,Q", [(8, 4096, 0, 4327, 1536, 1536), (3, 4032, 0, 4263, 4608, 1536), (2, 128, 64, 320, 256, 512), (4, 1024, 77, 1101, 640, 640),
                                             (8, 4096, 0, 4327, 256, 256)])
def test_gemm_tn_over_a_segmented_contraction_axis_is_bit_equal_to_the_gathered_copy(ops, B, rows, lo, S, P, Q):
    """st355_gemm_tn_seg_bf16: the rows [lo, lo + rows) of every sample of a joint [B * S, C] buffer contracted IN PLACE (either operand, or both) == st355_gemm_tn_bf16
    on the gathered compact copy, bit for bit (same tiles, same K-slices, same accumulation order; only the K-tile addresses differ)"""
    torch.manual_seed(63)
    d_ = dev()
    jl = (torch.randn(B * S, P, device=d_) * 0.5).to(BF16); jr = torch.randn(B * S, Q, device=d_).to(BF16)
    vl, vr = jl.view(B, S, P)[:, lo:lo + rows], jr.view(B, S, Q)[:, lo:lo + rows]
    cl, cr = vl.reshape(B * rows, P), vr.reshape(B * rows, Q)
    ref = ops.gemm_tn(cl, cr)
    assert rel(ref, cl.float().t() @ cr.float()) < 5e-3
    assert torch.equal(ops.gemm_tn(vl, cr), ref), "segmented L"
    assert torch.equal(ops.gemm_tn(cl, vr), ref), "segmented R"
    assert torch.equal(ops.gemm_tn(vl, vr), ref), "both segmented"
    acc0 = torch.randn(P, Q, device=d_).to(BF16)
    a1, a2 = acc0.clone(), acc0.clone()
    ops.gemm_tn(cl, cr, out=a1, accumulate=True); ops.gemm_tn(vl, cr, out=a2, accumulate=True)
    assert torch.equal(a1, a2)


@pytest.mark.parametrize("B,H,S,hd", [(1, 24, 4327, 64), (2, 3, 4183, 64), (2, 4, 231 + 64, 64), (1, 5, 1000, 64), (1, 2, 333, 96), (2, 2, 300, 128), (1, 2, 65, 64)])
def test_attention_dq_64_row_kernel_with_a_ragged_key_tail(ops, B, H, S, hd):
    """S % 64 != 0 (SD3: 4096 + 231; every mixed-aspect bucket): k_attn_bwd_dq64 takes the full 64-key tiles, the general kernel adds the last (ragged) tile to the dQ it
    left — the backward is separable over keys given lse2 and delta.  Against the 32-row kernel alone: equal up to ONE extra bf16 rounding of the sum (rel-L2 <= 3e-3,
    stated; measured ~1.5e-3), and against fp32 torch autograd within the bf16 bound of the suite's attention test (2e-2).  dK / dV are untouched by the dispatch."""
    torch.manual_seed(64)
    d_ = dev()
    D = H * hd
    dv_ = 72 if hd == 96 else hd
    scale = 1.0 / math.sqrt(dv_)
    mk = lambda *sh: torch.randn(*sh, device=d_)
    Q, K = mk(B, H, S, hd), mk(B, H, S, hd)
    Q[..., dv_:] = 0; K[..., dv_:] = 0
    Q, K = Q.to(BF16), K.to(BF16)
    V = mk(B * S, H, hd); V[..., dv_:] = 0
    V = V.reshape(B * S, D).to(BF16)
    Sp = (S + 63) // 64 * 64
    Vt = torch.zeros(B, H, hd, Sp, device=d_, dtype=BF16); Vt[..., :S] = V.view(B, S, H, hd).permute(0, 2, 3, 1)
    dO = mk(B * S, H, hd); dO[..., dv_:] = 0
    dO = dO.reshape(B * S, D).to(BF16)
    O = torch.empty(B * S, D, device=d_, dtype=BF16); lse2 = torch.empty(B, H, S, device=d_)
    ops.attn_fwd(Q, K, Vt, O, lse2, B, H, S, Sp, hd, scale)
    res = {}
    prev = ops.attn_set_impl()
    try:
        for impl in (32, 64):
            ops.attn_set_impl(dq=impl)
            dQ = torch.empty_like(Q); dK = torch.empty_like(K); dqkv = torch.zeros(B * S, 3 * D, device=d_, dtype=BF16)
            ops.attn_bwd(Q, K, None, None, V, O, dO, lse2, dQ, dK, dqkv[:, 2 * D:], B, H, S, Sp, hd, scale)
            res[impl] = (dQ, dK, dqkv)
    finally:
        ops.attn_set_impl(fwd=prev[0], dq=prev[1], dkv=prev[2])
    r = rel(res[64][0], res[32][0])
    print(f"[parity] dq64 + ragged tail vs dq (B{B} H{H} S{S} d{hd}): rel_l2={r:.3e}")
    assert r < 3e-3
    assert torch.equal(res[64][1], res[32][1]) and torch.equal(res[64][2], res[32][2])
    q, k = Q.float().requires_grad_(True), K.float().requires_grad_(True)
    v = V.float().view(B, S, H, hd).permute(0, 2, 1, 3)
    o = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1) @ v
    (o * dO.float().view(B, S, H, hd).permute(0, 2, 1, 3)).sum().backward()
    assert rel(res[64][0], q.grad) < 2e-2
    if dv_ < hd:
        assert res[64][0][..., 80:].abs().max().item() == 0


def test_ema_update_inside_the_adamw_launch_is_bit_equal_to_the_separate_pass(ops):
    """st355_adamw_ema_step_bf16 with the `ema` pointer == the same step followed by st355_ema_update (ema.py:393-433: (s - p) materialised in the parameter dtype)"""
    torch.manual_seed(65)
    d_ = dev()
    n = 8 * 100_003
    p0 = torch.randn(n, device=d_).to(BF16); g = (torch.randn(n, device=d_) * 0.1).to(BF16)
    s0 = (p0.float() + 0.01 * torch.randn(n, device=d_)).to(BF16)
    out = []
    for fused in (False, True):
        p, s = p0.clone(), s0.clone()
        m = torch.zeros(n, device=d_); v = torch.zeros(n, device=d_)
        for step in (1, 2, 3):
            ops.adamw_ema_step(p, g, m, v, step, 1e-3, grad_scale=0.5, ema=s if fused else None, ema_decay=0.999)
            if not fused:
                ops.ema_update(s, p, 0.999)
        out.append((p, s, m, v))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    assert not torch.equal(out[0][1], s0)


@pytest.mark.parametrize("M,K,F", [(4096, 1280, 5120), (1024, 640, 2560), (300, 320, 1280), (16384, 1280, 5120)])
def test_geglu_inside_the_feed_forward_gemms(ops, M, K, F):
    """ST355_EPI_GEGLU / ST355_EPI_GEGLU_GRAD on interleaved projection rows (ops.geglu_interleave) against the unfused sequence GEMM -> st355_geglu_fwd and
    GEMM -> st355_geglu_bwd -> GEMM: the forward output bit-identical (same bf16 pre-activation, same erf GELU), the kept pre-activation = the unfused one with its
    columns interleaved, the input gradient to bf16 rounding of d out (<= 4e-3 rel-L2: the fused epilogue rounds the same values at the same points; only the last
    GEMM's contraction ORDER over the 2F columns differs)."""
    torch.manual_seed(66)
    d_ = dev()
    x = torch.randn(M, K, device=d_).to(BF16)
    w1 = (torch.randn(2 * F, K, device=d_) / math.sqrt(K)).to(BF16); b1 = (torch.randn(2 * F, device=d_) * 0.1).to(BF16)
    w2 = (torch.randn(K, F, device=d_) / math.sqrt(F)).to(BF16)
    dy = torch.randn(M, K, device=d_).to(BF16)
    # unfused
    f = ops.gemm(x, w1, bias=b1)
    g0 = ops.geglu_fwd(f)
    dg = ops.gemm(dy, w2.t().contiguous())                     # d out [M, F] = dy W2   (W2^T as the NT weight operand)
    df = ops.geglu_bwd(f, dg)
    dx0 = ops.gemm(df, w1.t().contiguous())
    # fused
    w_il, b_il = ops.geglu_interleave(w1, b1)
    f_il = torch.empty(M, 2 * F, device=d_, dtype=BF16)
    g1 = ops.gemm(x, w_il, bias=b_il, epilogue=ops.EPI_GEGLU, aux_out=f_il)
    perm = torch.stack([torch.arange(F).view(F // 32, 32), F + torch.arange(F).view(F // 32, 32)], dim=1).reshape(-1).to(d_)
    df_il = ops.gemm(dy, w2.t().contiguous(), epilogue=ops.EPI_GEGLU_GRAD, aux_in=f_il)
    if M >= 1024:             # both forms on the 256x256 schedule: the same accumulators
        assert torch.equal(g1, g0) and torch.equal(f_il, f[:, perm]) and torch.equal(df_il, df[:, perm])
    else:                     # (a small problem's unfused GEMMs run on the 128x128 schedule: same formula, another tile walk)
        assert rel(g1, g0) < 2e-3 and rel(f_il, f[:, perm]) < 2e-3 and rel(df_il, df[:, perm]) < 4e-3
    dx1 = ops.gemm(df_il, ops.transpose(w_il))
    assert rel(dx1, dx0) < 4e-3
    ref = (torch.nn.functional.gelu(f.float()[:, F:]) * f.float()[:, :F])
    assert rel(g1, ref) < 4e-3


@pytest.mark.parametrize("M,K", [(16384, 1280), (65536, 640), (16384, 3840), (1024, 256), (5000, 704), (16421, 1280), (1061, 512), (4096, 1792)])
def test_thin_rank_space_gemm(ops, M, K):
    """k_gemm_rows (K < 2048, K % 256 == 0: rows split over the waves, operands staged through LDS — ragged last row blocks included) / k_gemm_thin (N = 64: the LoRA down projections x A^T and dY (sB)): one streaming pass over the activations, against fp32 torch (bf16 output rounding + the
    fp32 accumulation order of four interleaved K partials: rel-L2 <= 4e-3), its N = 128 form, and the tile path through an N = 256 problem whose first columns are
    the same products."""
    torch.manual_seed(67)
    d_ = dev()
    x = torch.randn(M, K, device=d_).to(BF16)
    w = (torch.randn(128, K, device=d_) / math.sqrt(K)).to(BF16)
    y = ops.gemm(x, w[:64])
    ref = x.float() @ w[:64].float().t()
    assert rel(y, ref) < 4e-3
    y128 = ops.gemm(x, w)                                   # N = 128: the single-buffer form of the same kernel (another accumulation order)
    assert rel(y128, x.float() @ w.float().t()) < 4e-3
    assert rel(y, y128[:, :64]) < 3e-3
    y256 = ops.gemm(x, torch.cat([w, w], 0))                # N = 256: the tile path
    assert rel(y128, y256[:, :128]) < 3e-3
    wide = torch.zeros(M, 192, device=d_, dtype=BF16)       # a strided destination (a column block of a wider buffer)
    ops.gemm(x, w[:64], out=wide[:, 64:128])
    assert torch.equal(wide[:, 64:128], y) and float(wide[:, :64].abs().max()) == 0 and float(wide[:, 128:].abs().max()) == 0


@pytest.mark.parametrize("B,R,S,pos0,H,K,parts,lora,with_bias,with_vt", [
    (2, 1024, 1024, 0, 20, 1280, "qkv", True, False, True),        # SDXL 32^2 level attn1 (LoRA on the fused projection)
    (1, 4096, 4096, 0, 10, 640, "qkv", False, True, True),         # SDXL 64^2 level
    (2, 512, 768, 256, 4, 256, "qkv", False, True, True),          # a stream's rows inside a joint sequence (pos0 > 0)
    (3, 231, 743, 512, 6, 384, "qkv", True, True, False),          # ragged rows per sample (SD3's 231 text rows): no V^T
    (2, 1024, 1024, 0, 20, 1280, "q", True, False, False),         # attn2: the query projection alone
    (2, 200, 200, 0, 2, 128, "qk", False, False, False),
])
def test_gemm_head_split_epilogue_is_bit_equal_to_projection_then_head_split(ops, B, R, S, pos0, H, K, parts, lora, with_bias, with_vt):
    """ST355_EPI_HEADS: head-major q / k, row-major v and head-major V^T straight from the projection GEMM's accumulators == st355_gemm_bf16 (EPI_NONE) followed by
    st355_head_split_pad, bit for bit (same accumulation, bias add and ONE bf16 rounding; only the store addresses differ)"""
    torch.manual_seed(65)
    d_ = dev()
    C_ = H * 64
    n_q = C_ if "q" in parts else 0
    n_k = C_ if "k" in parts else 0
    n_v = C_ if "v" in parts else 0
    N = n_q + n_k + n_v
    M = B * R
    x = torch.randn(M, K, device=d_).to(BF16)
    w = (torch.randn(N, K, device=d_) / math.sqrt(K)).to(BF16)
    bias = torch.randn(N, device=d_).to(BF16) if with_bias else None
    kw = {}
    if lora:
        kw = dict(a2=torch.randn(M, 64, device=d_).to(BF16), b2=(torch.randn(N, 64, device=d_) * 0.1).to(BF16))
    ref = ops.gemm(x, w, bias=bias, **kw)                                                       # [M, N]
    Sp = (S + 63) // 64 * 64
    mk = lambda: torch.full((B, H, S, 64), 7.0, device=d_, dtype=BF16)
    Q = mk() if n_q else None
    Kh = mk() if n_k else None
    Vt = torch.full((B, H, 64, Sp), 7.0, device=d_, dtype=BF16) if with_vt else None
    out = torch.full((M, n_v), 7.0, device=d_, dtype=BF16) if n_v else None
    okw = dict(out=out) if n_v else {}
    ops.gemm(x, w, bias=bias, epilogue=ops.EPI_HEADS, heads=ops.heads(Q, Kh, Vt, H, S, pos0, n_q, n_k), rows_per_batch=R, **okw, **kw)
    col = 0
    for n, dst, nm in ((n_q, Q, "q"), (n_k, Kh, "k")):
        if not n:
            continue
        want = ref[:, col:col + n].reshape(B, R, H, 64).permute(0, 2, 1, 3)
        assert torch.equal(dst[:, :, pos0:pos0 + R], want), nm
        rest = torch.cat([dst[:, :, :pos0].reshape(-1), dst[:, :, pos0 + R:].reshape(-1)])
        assert rest.numel() == 0 or (float(rest.float().min()) == 7.0 and float(rest.float().max()) == 7.0), nm + ": rows outside [pos0, pos0 + R) untouched"
        col += n
    if n_v:
        assert torch.equal(out, ref[:, col:]), "v rows"
        if with_vt:
            assert torch.equal(Vt[..., pos0:pos0 + R], ref[:, col:].reshape(B, R, H, 64).permute(0, 2, 3, 1)), "v^T"
            rest = torch.cat([Vt[..., :pos0].reshape(-1), Vt[..., pos0 + R:].reshape(-1)])
            assert rest.numel() == 0 or (float(rest.float().min()) == 7.0 and float(rest.float().max()) == 7.0)
    if parts == "qkv" and pos0 == 0 and R == S:                                                # and against the kernel it replaces
        Xq, _, _ = ops.head_split(ref[:, :C_], B, H, 64, S, want_xt=False)
        _, Xvt, _ = ops.head_split(ref[:, 2 * C_:], B, H, 64, S, want_x=False)
        assert torch.equal(Xq, Q)
        if with_vt:
            assert torch.equal(Xvt[..., :S], Vt[..., :S])


@pytest.mark.parametrize("M,N,K,epi,lora,with_bias", [
    (16384, 1280, 2560, "none", True, False),        # 320 tiles: one full round + 64 tiles cut in two (K + K2 < 4096), the adapter's K-extension on the last slice
    (16384, 1280, 5120, "add", False, True),         # ... cut in four: the SDXL 32^2 level's feed-forward down projection at batch 16
    (16384, 1280, 10240, "add", True, True),
    (16000, 1280, 2048, "gelu", False, True),        # ragged last row tile: 315 tiles, 59 cut (the cut count is not a multiple of the 8 XCDs)
    (16384, 1280, 1280, "none", True, False),        # short contraction: not cut (bit-equal)
    (8192, 1280, 2560, "none", False, False),        # 160 tiles: below one round, more than half the chip -> not cut
    (34816, 1536, 6144, "gate_residual", False, True),   # 816 tiles = 3 rounds + 48 tiles cut in four (SD3-Medium rows)
    (32768, 256, 2048, "none", False, False),        # 128 tiles, no full round: every tile cut in two
])
def test_gemm_stream_k_tail_matches_the_uncut_schedule(ops, M, N, K, epi, lora, with_bias):
    """the stream-K tail (k_gemm_pq<..., SK>): the tiles of a mostly empty last round cut along K over the idle CUs.  Against the uncut schedule (set_tail_split(0)):
    the same products summed in fp32 with a different association — equal up to the bf16 rounding of the output (rel-L2 <= 2e-3 stated, measured ~1e-4; no element
    more than 2 bf16 ulps apart where the value is not tiny), against fp32 torch within the suite's GEMM bound (5e-3); repeated launches are bit-identical (fixed
    slice order, counters left at zero)."""
    torch.manual_seed(66)
    d_ = dev()
    x = torch.randn(M, K, device=d_).to(BF16)
    w = (torch.randn(N, K, device=d_) / math.sqrt(K)).to(BF16)
    kw = {}
    if with_bias:
        kw["bias"] = torch.randn(N, device=d_).to(BF16)
    if lora:
        kw.update(a2=torch.randn(M, 64, device=d_).to(BF16), b2=(torch.randn(N, 64, device=d_) * 0.1).to(BF16))
    ref = x.float() @ w.float().t()
    if lora:
        ref = ref + kw["a2"].float() @ kw["b2"].float().t()
    if with_bias:
        ref = ref + kw["bias"].float()
    if epi == "add":
        res = torch.randn(M, N, device=d_).to(BF16)
        kw.update(epilogue=ops.EPI_ADD, aux_in=res)
        ref = ref + res.float()
    elif epi == "gelu":
        kw.update(epilogue=ops.EPI_GELU)
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    elif epi == "gate_residual":
        rpb = M // 8
        res = torch.randn(M, N, device=d_).to(BF16); gate = torch.randn(8, N, device=d_).to(BF16)
        kw.update(epilogue=ops.EPI_GATE_RESIDUAL, aux_in=res, gate=gate, rows_per_batch=rpb)
        ref = res.float() + gate.float().repeat_interleave(rpb, dim=0) * ref
    prev = ops.gemm_set_tail_split(0)
    try:
        y0 = ops.gemm(x, w, **kw)
        ops.gemm_set_tail_split(1)
        y1 = ops.gemm(x, w, **kw)
        for _ in range(3):
            assert torch.equal(ops.gemm(x, w, **kw), y1), "repeated launches"
    finally:
        ops.gemm_set_tail_split(prev)
    assert ops.gemm_tail_placement() in (0, 1), "the XCD placement probe refused the stream-K tail on this device"      # (0: no launch of this process was cut yet)
    r = rel(y1, y0)
    print(f"[parity] stream-K tail vs uncut {M}x{N}x{K} {epi}: rel_l2={r:.3e} vs fp32 {rel(y1, ref):.3e} (uncut vs fp32 {rel(y0, ref):.3e})")
    assert r < 2e-3
    big = y0.float().abs() > 0.05
    assert float(((y1.float() - y0.float()).abs() / y0.float().abs().clamp_min(0.05))[big].max()) <= 2 ** -6
    assert rel(y1, ref) < 5e-3
