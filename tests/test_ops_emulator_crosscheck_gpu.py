"""Pins tests/ops_emulator.py to the kernels it stands in for: the emulated wrappers on random operands, the emulator on the CPU against libst355 on the MI355X
(bf16 rounding apart) — GEMM epilogues, the TN GEMM, token-axis reductions, AdaLN forward / backward, RMSNorm + RoPE forward / backward with the norm-weight gradients,
attention forward / backward with a key bias, the UNet's grid-buffer ops (layout, GroupNorm, 3x3 convolution and its weight gradient, upsampling), the fused loss and AdamW.
First run: round 3, `profiles/archive/r03t_emulator_crosscheck.log` (5 passed).  The CPU host-sequencing tests then carry the kernels' semantics, not just the emulator's."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
DEV = "cuda:0"


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _bf(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=_g(seed)) * scale).to(BF16)


def _close(a, b, tol=2e-2):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item() < tol


def _both(fn_name, make_args):
    """run ops.<fn> on device tensors and ops_emulator.<fn> on the same host tensors; returns (device result, host result)"""
    from simpletuner_amd import ops
    from tests import ops_emulator as EMU
    host = make_args()
    dev = [t.to(DEV) if torch.is_tensor(t) else t for t in host]
    return getattr(ops, fn_name)(*dev), getattr(EMU, fn_name)(*host), dev, host


def test_gemm_epilogues_and_tn():
    from simpletuner_amd import ops
    from tests import ops_emulator as EMU
    a, w, b, x = _bf(512, 256, seed=1), _bf(384, 256, seed=2, scale=0.06), _bf(384, seed=3, scale=0.1), _bf(512, 384, seed=4)
    gate = _bf(2, 384, seed=5)
    for epi, kw in ((ops.EPI_NONE, {}), (ops.EPI_ADD, dict(aux_in=x)), (ops.EPI_GELU, {}), (ops.EPI_MUL_GELU_GRAD, dict(aux_in=x)),
                    (ops.EPI_GATE_RESIDUAL, dict(aux_in=x, gate=gate, rows_per_batch=256))):
        d = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), epilogue=epi, **{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()})
        h = EMU.gemm(a, w, bias=b, epilogue=epi, **kw)
        assert _close(d, h, 1e-2), epi
    L, R = _bf(256, 128, seed=6), _bf(256, 192, seed=7)
    assert _close(ops.gemm_tn(L.to(DEV), R.to(DEV)), EMU.gemm_tn(L, R), 1e-2)
    out_d, out_h = torch.zeros(2, 384, dtype=F32, device=DEV), torch.zeros(2, 384, dtype=F32)
    ops.colsum_prod(x.to(DEV), out_d, b=_bf(512, 384, seed=8).to(DEV), rows_per_batch=256)
    EMU.colsum_prod(x, out_h, b=_bf(512, 384, seed=8), rows_per_batch=256)
    assert _close(out_d, out_h, 1e-2)


def test_adaln_and_qk_norm_rope():
    from simpletuner_amd import ops
    from tests import ops_emulator as EMU
    x, sc, sh, dy = _bf(128, 256, seed=1), _bf(2, 256, seed=2, scale=0.3), _bf(2, 256, seed=3, scale=0.3), _bf(128, 256, seed=4)
    assert _close(ops.ln_modulate_fwd(x.to(DEV), sc.to(DEV), sh.to(DEV), 64), EMU.ln_modulate_fwd(x, sc, sh, 64), 1e-2)
    d = ops.ln_modulate_bwd(dy.to(DEV), x.to(DEV), sc.to(DEV), 64, dres=x.to(DEV), gate=sh.to(DEV), want_gated=True)
    h = EMU.ln_modulate_bwd(dy, x, sc, 64, dres=x, gate=sh, want_gated=True)
    assert _close(d[0], h[0], 1e-2) and _close(d[1], h[1], 1e-2)
    B, H, hd, S = 2, 2, 128, 96
    qkv, wq, wk = _bf(B * S, 3 * H * hd, seed=5), (1 + 0.2 * torch.randn(hd, generator=_g(6))).to(BF16), (1 + 0.2 * torch.randn(hd, generator=_g(7))).to(BF16)
    ang = torch.rand(S, hd // 2, generator=_g(8)) * 6.28
    cos, sin = ang.cos().repeat_interleave(2, 1).contiguous(), ang.sin().repeat_interleave(2, 1).contiguous()
    Sp = 128
    outs = []
    for mod, dev in ((ops, DEV), (EMU, "cpu")):
        Q = torch.zeros(B, H, S, hd, dtype=BF16, device=dev); K = torch.zeros_like(Q); Vt = torch.zeros(B, H, hd, Sp, dtype=BF16, device=dev)
        mod.qk_norm_rope_fwd(qkv.to(dev), wq.to(dev), wk.to(dev), cos.to(dev), sin.to(dev), Q, K, None, None, Vt, B, H, hd, S, 0, S, Sp)
        dQ, dK = _bf(B, H, S, hd, seed=9).to(dev), _bf(B, H, S, hd, seed=10).to(dev)
        dqkv = torch.zeros(B * S, 3 * H * hd, dtype=BF16, device=dev)
        gq, gk = torch.zeros(hd, dtype=BF16, device=dev), torch.zeros(hd, dtype=BF16, device=dev)
        mod.qk_norm_rope_bwd_wgrad(dQ, dK, qkv.to(dev), wq.to(dev), wk.to(dev), cos.to(dev), sin.to(dev), dqkv, B, H, hd, S, 0, S, gq, gk)
        outs.append((Q, K, Vt, dqkv, gq, gk))
    for d_, h_ in zip(*outs):
        assert _close(d_, h_, 1.5e-2)


def test_attention_forward_and_backward():
    from simpletuner_amd import ops
    from tests import ops_emulator as EMU
    B, H, S, hd, Sp = 2, 2, 160, 128, 192
    Q, K, V = _bf(B, H, S, hd, seed=1, scale=0.5), _bf(B, H, S, hd, seed=2, scale=0.5), _bf(B * S, H * hd, seed=3)
    kb = torch.zeros(B, S); kb[0, 100:] = -3.0
    res = []
    for mod, dev in ((ops, DEV), (EMU, "cpu")):
        Vt = torch.zeros(B, H, hd, Sp, dtype=BF16, device=dev)
        Vt[..., :S] = V.to(dev).reshape(B, S, H, hd).permute(0, 2, 3, 1)
        O = torch.zeros(B * S, H * hd, dtype=BF16, device=dev); lse2 = torch.zeros(B, H, S, dtype=F32, device=dev)
        mod.attn_fwd(Q.to(dev), K.to(dev), Vt, O, lse2, B, H, S, Sp, hd, hd ** -0.5, key_bias=kb.to(dev))
        dO = _bf(B * S, H * hd, seed=4).to(dev)
        dQ, dK = torch.zeros(B, H, S, hd, dtype=BF16, device=dev), torch.zeros(B, H, S, hd, dtype=BF16, device=dev)
        dV = torch.zeros(B * S, H * hd, dtype=BF16, device=dev)
        mod.attn_bwd(Q.to(dev), K.to(dev), None, None, V.to(dev), O, dO, lse2, dQ, dK, dV, B, H, S, Sp, hd, hd ** -0.5, key_bias=kb.to(dev))
        res.append((O, lse2, dQ, dK, dV))
    for d_, h_ in zip(*res):
        assert _close(d_, h_, 2e-2)


def test_grid_ops_of_the_unet_path():
    from simpletuner_amd import ops
    from tests import ops_emulator as EMU
    B, H, W, Cin, Cout = 2, 8, 12, 64, 128
    x = _bf(B, Cin, H, W, seed=1)
    w, b = _bf(Cout, 9 * Cin, seed=2, scale=0.04), _bf(Cout, seed=3, scale=0.1)
    gam, bet = (1 + 0.1 * torch.randn(Cin, generator=_g(4))).to(BF16), _bf(Cin, seed=5, scale=0.1)
    res = []
    for mod, dev in ((ops, DEV), (EMU, "cpu")):
        g = mod.grid_from_nchw(x.to(dev), Cin)
        y, stats = mod.groupnorm_fwd(g, gam.to(dev), bet.to(dev), B, H, W, groups=32, eps=1e-5, silu=True)
        c = mod.conv(y, w.to(dev), B, H, W, bias=b.to(dev), taps=9)
        dw = torch.zeros(Cout, 9 * Cin, dtype=BF16, device=dev)
        mod.conv_wgrad(y, c, dw, B, H, W, taps=9)
        dx = mod.groupnorm_bwd(y, g, gam.to(dev), bet.to(dev), stats, B, H, W, groups=32, silu=True)
        up = mod.upsample2x(g, B, H, W)
        res.append((mod.grid_to_nchw(c, B, Cout, H, W), dw, mod.grid_to_nchw(dx, B, Cin, H, W), mod.grid_to_nchw(up, B, Cin, 2 * H, 2 * W), stats))
    for d_, h_ in zip(*res):
        assert _close(d_, h_, 2e-2)


def test_loss_and_optimizer():
    from simpletuner_amd import ops
    from tests import ops_emulator as EMU
    p, t, w = _bf(4, 16, 8, 8, seed=1), _bf(4, 16, 8, 8, seed=2), torch.rand(4, generator=_g(3)) + 0.5
    d = ops.mse_loss(p.to(DEV), t.to(DEV), weight=w.to(DEV))
    h = EMU.mse_loss(p, t, weight=w)
    assert all(_close(a, b, 1e-2) for a, b in zip(d, h))
    n = 4096
    P, G = torch.randn(n, generator=_g(4)), torch.randn(n, generator=_g(5)) * 0.1
    state = []
    for mod, dev in ((ops, DEV), (EMU, "cpu")):
        pp, gg, m, v = P.clone().to(dev), G.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        for step in (1, 2, 3):
            mod.adamw_ema_step(pp, gg, m, v, step, 1e-2, 0.9, 0.999, 1e-8, 1e-2, grad_scale=0.5)
        state.append((pp, m, v))
    for d_, h_ in zip(*state):
        assert _close(d_, h_, 1e-4)
