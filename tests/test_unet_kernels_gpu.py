"""GPU parity of the UNet-path kernels (conv-as-GEMM over zero-bordered grids, GroupNorm, GEGLU, affine LayerNorm, cross-attention)
against plain PyTorch fp32 references of the same ops (F.conv2d / F.group_norm / F.layer_norm / softmax attention + autograd).
Tolerances: bf16 in/out with fp32 accumulation -> rel-L2 <= 1e-2 on outputs and gradients (stated per assert)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _ohwi(w):          # torch Conv2d weight [O,I,kh,kw] -> [O, kh*kw*I]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 24, 64, 64), (1, 9, 7, 128, 72), (4, 64, 64, 320, 640)])
def test_conv3x3_grid_vs_torch(B, H, W, Cin, Cout):
    from simpletuner_amd import ops
    dev = "cuda:0"
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W, device=dev).to(BF16)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(9 * Cin)).to(BF16)
    b = torch.randn(Cout, device=dev).to(BF16)
    temb = torch.randn(B, Cout, device=dev).to(BF16)
    res = torch.randn(B, Cout, H, W, device=dev).to(BF16)
    xg = ops.grid_from_nchw(x, Cin)
    rg = ops.grid_from_nchw(res, Cout)
    out = ops.conv(xg, _ohwi(w), B, H, W, bias=b, img_add=temb, residual=rg)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + temb.float()[:, :, None, None] + res.float()
    got = ops.grid_to_nchw(out, B, Cout, H, W)
    assert _rel(got, ref) < 6e-3, _rel(got, ref)
    # the border of the output grid and its tail rows are exactly zero (the next conv relies on it)
    g4 = out[:B * (H + 2) * (W + 2)].view(B, H + 2, W + 2, Cout)
    assert g4[:, 0].abs().max() == 0 and g4[:, -1].abs().max() == 0 and g4[:, :, 0].abs().max() == 0 and g4[:, :, -1].abs().max() == 0
    assert out[B * (H + 2) * (W + 2):].abs().max() == 0


def test_conv1x1_and_wgrad_vs_autograd():
    from simpletuner_amd import ops
    dev = "cuda:0"
    torch.manual_seed(1)
    B, H, W, Cin, Cout = 2, 20, 12, 128, 64
    x = torch.randn(B, Cin, H, W, device=dev).to(BF16)
    w3 = (torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(9 * Cin)).to(BF16)
    w1 = (torch.randn(Cout, Cin, 1, 1, device=dev) / math.sqrt(Cin)).to(BF16)
    dy = torch.randn(B, Cout, H, W, device=dev).to(BF16)
    xg, dyg = ops.grid_from_nchw(x, Cin), ops.grid_from_nchw(dy, Cout)
    # 1x1
    o1 = ops.grid_to_nchw(ops.conv(xg, _ohwi(w1), B, H, W, taps=1), B, Cout, H, W)
    assert _rel(o1, F.conv2d(x.float(), w1.float())) < 6e-3
    # weight gradients (3x3 and 1x1) and input gradient (conv with flipped / transposed taps)
    xf = x.float().requires_grad_(True)
    w3f, w1f = w3.float().requires_grad_(True), w1.float().requires_grad_(True)
    (F.conv2d(xf, w3f, padding=1) * dy.float()).sum().backward()
    dw3 = torch.empty(Cout, 9 * Cin, dtype=BF16, device=dev)
    ops.conv_wgrad(xg, dyg, dw3, B, H, W, taps=9)
    assert _rel(dw3, _ohwi(w3f.grad)) < 8e-3, _rel(dw3, _ohwi(w3f.grad))
    wt = w3.view(Cout, Cin, 9).flip(2).permute(1, 2, 0).reshape(Cin, 9 * Cout).contiguous()     # [ci, (8-tap)*Cout + co]
    dx = ops.grid_to_nchw(ops.conv(dyg, wt, B, H, W), B, Cin, H, W)
    assert _rel(dx, xf.grad) < 8e-3, _rel(dx, xf.grad)
    (F.conv2d(x.float(), w1f) * dy.float()).sum().backward()
    dw1 = torch.empty(Cout, Cin, dtype=BF16, device=dev)
    ops.conv_wgrad(xg, dyg, dw1, B, H, W, taps=1)
    assert _rel(dw1, w1f.grad.view(Cout, Cin)) < 8e-3


def test_downsample_conv_via_columns_and_upsample():
    from simpletuner_amd import ops
    dev = "cuda:0"
    torch.manual_seed(2)
    B, H, W, Cn, Cout = 2, 16, 24, 64, 128
    x = torch.randn(B, Cn, H, W, device=dev).to(BF16)
    w = (torch.randn(Cout, Cn, 3, 3, device=dev) / math.sqrt(9 * Cn)).to(BF16)
    xg = ops.grid_from_nchw(x, Cn)
    col = ops.im2col3x3(xg, B, H, W, stride=2)
    assert col.shape[1] == 9 * Cn
    out = ops.grid_to_nchw(ops.conv(col, _ohwi(w), B, H // 2, W // 2, taps=1), B, Cout, H // 2, W // 2)
    xf = x.float().requires_grad_(True)
    ref = F.conv2d(xf, w.float(), stride=2, padding=1)
    assert _rel(out, ref) < 6e-3
    dy = torch.randn_like(ref).to(BF16)
    (ref * dy.float()).sum().backward()
    dyg = ops.grid_from_nchw(dy, Cout)
    n_in = B * (H // 2 + 2) * (W // 2 + 2)
    dcol = torch.zeros_like(col)
    ops.gemm(dyg[:n_in], _ohwi(w).t().contiguous(), out=dcol[:n_in])
    dx = ops.grid_to_nchw(ops.col2im3x3(dcol, B, H, W, Cn, stride=2), B, Cn, H, W)
    assert _rel(dx, xf.grad) < 8e-3, _rel(dx, xf.grad)
    # tiny-channel column path (conv_in: 4 latent channels padded to 8, K = 72 -> 128)
    lat = torch.randn(B, 4, H, W, device=dev).to(BF16)
    w_in = (torch.randn(64, 4, 3, 3, device=dev) / 6).to(BF16)
    lg = ops.grid_from_nchw(lat, 8)
    c8 = ops.im2col3x3(lg, B, H, W, stride=1)
    w8 = torch.zeros(64, c8.shape[1], dtype=BF16, device=dev)
    w8.view(64, -1)[:, :72].view(64, 9, 8)[:, :, :4] = w_in.permute(0, 2, 3, 1).reshape(64, 9, 4)
    o_in = ops.grid_to_nchw(ops.conv(c8, w8, B, H, W, taps=1), B, 64, H, W)
    assert _rel(o_in, F.conv2d(lat.float(), w_in.float(), padding=1)) < 6e-3
    # nearest 2x upsample and its adjoint
    up = ops.upsample2x(xg, B, H, W)
    assert torch.equal(ops.grid_to_nchw(up, B, Cn, 2 * H, 2 * W), F.interpolate(x.float(), scale_factor=2.0, mode="nearest").to(BF16))
    g2 = torch.randn(B, Cn, 2 * H, 2 * W, device=dev).to(BF16)
    dn = ops.grid_to_nchw(ops.upsample2x_bwd(ops.grid_from_nchw(g2, Cn), B, H, W), B, Cn, H, W)
    assert _rel(dn, F.avg_pool2d(g2.float(), 2) * 4) < 4e-3


@pytest.mark.parametrize("B,H,W,Cn,silu,tokens", [(2, 16, 24, 320, True, False), (3, 8, 8, 2560, True, False), (2, 12, 20, 640, False, True)])
def test_groupnorm_fwd_bwd_vs_torch(B, H, W, Cn, silu, tokens):
    from simpletuner_amd import ops
    dev = "cuda:0"
    torch.manual_seed(3)
    x = (torch.randn(B, Cn, H, W, device=dev) * 1.7 + 0.4).to(BF16)
    gamma = (1 + 0.2 * torch.randn(Cn, device=dev)).to(BF16)
    beta = (0.1 * torch.randn(Cn, device=dev)).to(BF16)
    eps = 1e-6 if tokens else 1e-5
    xg = ops.grid_from_nchw(x, Cn)
    y, stats = ops.groupnorm_fwd(xg, gamma, beta, B, H, W, eps=eps, silu=silu, out_tokens=tokens)
    xf = x.float().requires_grad_(True)
    gf, bf_ = gamma.float().requires_grad_(True), beta.float().requires_grad_(True)
    ref = F.group_norm(xf, 32, gf, bf_, eps)
    if silu:
        ref = F.silu(ref)
    got = y.view(B, H, W, Cn).permute(0, 3, 1, 2) if tokens else ops.grid_to_nchw(y, B, Cn, H, W)
    assert _rel(got, ref) < 5e-3, _rel(got, ref)
    dy = torch.randn(B, Cn, H, W, device=dev).to(BF16)
    dadd = torch.randn(B, Cn, H, W, device=dev).to(BF16)
    (ref * dy.float()).sum().backward()
    dyt = dy.permute(0, 2, 3, 1).reshape(B * H * W, Cn).contiguous() if tokens else ops.grid_from_nchw(dy, Cn)
    dgam = torch.empty(Cn, dtype=torch.float32, device=dev)
    dbet = torch.empty(Cn, dtype=torch.float32, device=dev)
    dx = ops.groupnorm_bwd(dyt, xg, gamma, beta, stats, B, H, W, silu=silu, dy_tokens=tokens, dadd=ops.grid_from_nchw(dadd, Cn), dgamma=dgam, dbeta=dbet)
    dxn = ops.grid_to_nchw(dx, B, Cn, H, W)
    assert _rel(dxn, xf.grad + dadd.float()) < 8e-3, _rel(dxn, xf.grad + dadd.float())
    assert _rel(dgam, gf.grad) < 5e-3 and _rel(dbet, bf_.grad) < 5e-3
    g4 = dx[:B * (H + 2) * (W + 2)].view(B, H + 2, W + 2, Cn)
    assert g4[:, 0].abs().max() == 0 and g4[:, :, -1].abs().max() == 0


def test_geglu_and_affine_layernorm_vs_torch():
    from simpletuner_amd import ops
    dev = "cuda:0"
    torch.manual_seed(4)
    M, D = 1000, 640
    h = torch.randn(M, 8 * D, device=dev).to(BF16)
    hf = h.float().requires_grad_(True)
    v, g = hf.chunk(2, dim=-1)
    ref = v * F.gelu(g)
    out = ops.geglu_fwd(h)
    assert _rel(out, ref) < 4e-3
    do = torch.randn(M, 4 * D, device=dev).to(BF16)
    (ref * do.float()).sum().backward()
    assert _rel(ops.geglu_bwd(h, do), hf.grad) < 6e-3
    x = (torch.randn(M, D, device=dev) * 2 + 0.3).to(BF16)
    wt = (1 + 0.2 * torch.randn(D, device=dev)).to(BF16)
    bs = (0.1 * torch.randn(D, device=dev)).to(BF16)
    xf, wf, bf_ = x.float().requires_grad_(True), wt.float().requires_grad_(True), bs.float().requires_grad_(True)
    ref = F.layer_norm(xf, (D,), wf, bf_, 1e-5)
    assert _rel(ops.layernorm_fwd(x, wt, bs), ref) < 4e-3
    dy = torch.randn(M, D, device=dev).to(BF16)
    dres = torch.randn(M, D, device=dev).to(BF16)
    (ref * dy.float()).sum().backward()
    assert _rel(ops.layernorm_bwd(dy, x, wt, dres=dres), xf.grad + dres.float()) < 6e-3
    dw = torch.empty(D, dtype=torch.float32, device=dev)
    db = torch.empty(D, dtype=torch.float32, device=dev)
    ops.layernorm_param_grads(dy, x, dw, db)
    assert _rel(dw, wf.grad) < 3e-3 and _rel(db, bf_.grad) < 3e-3


@pytest.mark.parametrize("B,H,Sq,Sk", [(2, 5, 1000, 77), (1, 10, 256, 300), (2, 4, 333, 333)])
def test_cross_attention_fwd_bwd_vs_torch(B, H, Sq, Sk):
    from simpletuner_amd import ops
    dev = "cuda:0"
    d = 64
    torch.manual_seed(5)
    Cm = H * d
    q = torch.randn(B * Sq, Cm, device=dev).to(BF16)
    kv = torch.randn(B * Sk, 2 * Cm, device=dev).to(BF16)
    scale = 1.0 / math.sqrt(d)
    Q, Qt, Sqp = ops.head_split(q, B, H, d, Sq)
    K, Kt, Skp = ops.head_split(kv[:, :Cm], B, H, d, Sk)
    _, Vt, _ = ops.head_split(kv[:, Cm:], B, H, d, Sk, want_x=False)
    O = torch.empty(B * Sq, Cm, dtype=BF16, device=dev)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=dev)
    ops.attn_cross_fwd(Q, K, Vt, O, lse, B, H, Sq, Sk, Skp, d, scale)
    qf = q.float().requires_grad_(True)
    kvf = kv.float().requires_grad_(True)
    qh = qf.view(B, Sq, H, d).transpose(1, 2)
    kh = kvf[:, :Cm].reshape(B, Sk, H, d).transpose(1, 2)
    vh = kvf[:, Cm:].reshape(B, Sk, H, d).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(B * Sq, Cm)
    assert _rel(O, ref) < 6e-3, _rel(O, ref)
    dO = torch.randn(B * Sq, Cm, device=dev).to(BF16)
    (ref * dO.float()).sum().backward()
    dQ = torch.empty_like(Q)
    dK = torch.empty_like(K)
    dkv = torch.zeros_like(kv)
    ops.attn_cross_bwd(Q, K, Qt, Kt, kv[:, Cm:], O, dO, lse, dQ, dK, dkv[:, Cm:], B, H, Sq, Sqp, Sk, Skp, d, scale)
    # the same backward without the head-major Q^T / K^T / dO^T copies (transposing LDS reads, r3: head_dim 64 / 96 too): bit-identical
    dQ2, dK2, dkv2 = torch.empty_like(Q), torch.empty_like(K), torch.zeros_like(kv)
    ops.attn_cross_bwd(Q, K, None, None, kv[:, Cm:], O, dO, lse, dQ2, dK2, dkv2[:, Cm:], B, H, Sq, Sqp, Sk, Skp, d, scale)
    # dQ: bit-identical (k_attn_bwd_dq64<64> where Sk % 64 == 0, else the 32-row kernel).  dK / dV without copies and without a key bias run k_attn_bwd_dkv4<64>
    # (r4: statistics folded into the MFMA chains — another fp32 summation order); the 32-key kernel it replaces stays bit-identical to the copy-reading one
    if Sk % 64 == 0 or Sk < 64:
        assert torch.equal(dQ2, dQ)
    else:                     # ragged key tail: dq64 + the general kernel on the last tile (r6): one extra bf16 rounding of the sum
        assert _rel(dQ2, dQ) < 3e-3
    assert _rel(dK2, dK) < 2e-3 and _rel(dkv2[:, Cm:], dkv[:, Cm:]) < 2e-3
    prev = ops.attn_set_impl(dkv=3)
    try:
        dQ3, dK3, dkv3 = torch.empty_like(Q), torch.empty_like(K), torch.zeros_like(kv)
        ops.attn_cross_bwd(Q, K, None, None, kv[:, Cm:], O, dO, lse, dQ3, dK3, dkv3[:, Cm:], B, H, Sq, Sqp, Sk, Skp, d, scale)
    finally:
        ops.attn_set_impl(dkv=prev[2])
    assert torch.equal(dK3, dK) and torch.equal(dkv3[:, Cm:], dkv[:, Cm:])
    for dQx, dKx, dkvx in ((dQ, dK, dkv), (dQ2, dK2, dkv2)):
        dq = torch.empty_like(q)
        ops.head_merge(dQx, dq, B, H, d, Sq)
        ops.head_merge(dKx, dkvx[:, :Cm], B, H, d, Sk)
        assert _rel(dq, qf.grad) < 1.2e-2, _rel(dq, qf.grad)
        assert _rel(dkvx, kvf.grad) < 1.2e-2, _rel(dkvx, kvf.grad)


@pytest.mark.parametrize("B,H,Sq,Sk,cross", [(2, 4, 700, 700, False), (1, 3, 333, 300, True), (2, 2, 1024, 77, True)])
def test_attention_head_dim_96_padded_72(B, H, Sq, Sk, cross):
    """PixArt's head_dim 72 runs zero-padded to 96 (scale 1/sqrt(72)): self- and cross-attention forward / backward vs torch on the 72 real dims"""
    from simpletuner_amd import ops
    dev = "cuda:0"
    d, dr = 96, 72
    torch.manual_seed(7)
    Cm = H * d
    pad = torch.zeros(1, H, d, device=dev); pad[:, :, :dr] = 1
    q = (torch.randn(B * Sq, H, d, device=dev) * pad).reshape(B * Sq, Cm).to(BF16)
    kv = (torch.randn(B * Sk, 2, H, d, device=dev) * pad[:, None]).reshape(B * Sk, 2 * Cm).to(BF16)
    kb = None
    if cross:
        kb = torch.zeros(B, Sk, device=dev); kb[:, Sk - 40:] = -10000.0
    scale = 1.0 / math.sqrt(dr)
    Q, Qt, Sqp = ops.head_split(q, B, H, d, Sq)
    K, Kt, Skp = ops.head_split(kv[:, :Cm], B, H, d, Sk)
    _, Vt, _ = ops.head_split(kv[:, Cm:], B, H, d, Sk, want_x=False)
    O = torch.empty(B * Sq, Cm, dtype=BF16, device=dev)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=dev)
    if cross:
        ops.attn_cross_fwd(Q, K, Vt, O, lse, B, H, Sq, Sk, Skp, d, scale, key_bias=kb)
    else:
        ops.attn_fwd(Q, K, Vt, O, lse, B, H, Sq, Sqp, d, scale)
    qf, kvf = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    qh = qf.view(B, Sq, H, d).transpose(1, 2)
    kh = kvf[:, :Cm].reshape(B, Sk, H, d).transpose(1, 2)
    vh = kvf[:, Cm:].reshape(B, Sk, H, d).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if kb is not None:
        s = s + kb[:, None, None, :]
    ref = (torch.softmax(s, dim=-1) @ vh).transpose(1, 2).reshape(B * Sq, Cm)
    assert _rel(O, ref) < 6e-3, _rel(O, ref)
    dO = (torch.randn(B * Sq, H, d, device=dev) * pad).reshape(B * Sq, Cm).to(BF16)
    (ref * dO.float()).sum().backward()
    dQ, dK = torch.empty_like(Q), torch.empty_like(K)
    dkv = torch.zeros_like(kv)
    if cross:
        ops.attn_cross_bwd(Q, K, Qt, Kt, kv[:, Cm:], O, dO, lse, dQ, dK, dkv[:, Cm:], B, H, Sq, Sqp, Sk, Skp, d, scale, key_bias=kb)
    else:
        ops.attn_bwd(Q, K, Qt, Kt, kv[:, Cm:], O, dO, lse, dQ, dK, dkv[:, Cm:], B, H, Sq, Sqp, d, scale)
    dQ2, dK2, dkv2 = torch.empty_like(Q), torch.empty_like(K), torch.zeros_like(kv)
    if cross:
        ops.attn_cross_bwd(Q, K, None, None, kv[:, Cm:], O, dO, lse, dQ2, dK2, dkv2[:, Cm:], B, H, Sq, Sqp, Sk, Skp, d, scale, key_bias=kb)
    else:
        ops.attn_bwd(Q, K, None, None, kv[:, Cm:], O, dO, lse, dQ2, dK2, dkv2[:, Cm:], B, H, Sq, Sqp, d, scale)
    if Sk % 64 == 0 or Sk < 64 or kb is not None:
        assert torch.equal(dQ2, dQ)                                                                        # no-copies form: bit-identical
    else:                     # ragged key tail: dq64 + the general kernel on the last tile (r6): one extra bf16 rounding of the sum
        assert _rel(dQ2, dQ) < 3e-3
    if cross:
        assert torch.equal(dK2, dK) and torch.equal(dkv2[:, Cm:], dkv[:, Cm:])
    else:
        # self-attention at head_dim 96 without copies runs k_attn_bwd_dkv4<96> (statistics folded into the MFMA chains: fp32 summation order differs); the
        # 32-key kernel it replaces stays bit-identical to the copy-reading one
        assert _rel(dK2, dK) < 2e-3 and _rel(dkv2[:, Cm:], dkv[:, Cm:]) < 2e-3
        prev = ops.attn_set_impl(dkv=3)
        try:
            dQ3, dK3, dkv3 = torch.empty_like(Q), torch.empty_like(K), torch.zeros_like(kv)
            ops.attn_bwd(Q, K, None, None, kv[:, Cm:], O, dO, lse, dQ3, dK3, dkv3[:, Cm:], B, H, Sq, Sqp, d, scale)
        finally:
            ops.attn_set_impl(dkv=prev[2])
        assert torch.equal(dK3, dK) and torch.equal(dkv3[:, Cm:], dkv[:, Cm:])
    for dQx, dKx, dkvx in ((dQ, dK, dkv), (dQ2, dK2, dkv2)):
        dq = torch.empty_like(q)
        ops.head_merge(dQx, dq, B, H, d, Sq)
        ops.head_merge(dKx, dkvx[:, :Cm], B, H, d, Sk)
        assert _rel(dq, qf.grad) < 1.2e-2, _rel(dq, qf.grad)
        assert _rel(dkvx, kvf.grad) < 1.2e-2, _rel(dkvx, kvf.grad)
