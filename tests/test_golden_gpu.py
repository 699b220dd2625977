"""HIP kernels (through the C ABI) vs tensors produced by the REFERENCE'S OWN CODE (tests/golden/reference_vectors.pt, see
tools/gen_golden.py).  Index/byte work is bit-exact; floating point within the stated tolerance."""
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
G = torch.load(ROOT / "tests" / "golden" / "reference_vectors.pt", weights_only=False)
DEV = "cuda:0"


def test_pack_unpack_vs_reference_bit_exact():
    from simpletuner_amd import ops

    packed = ops.flux_pack(G["pack.in"].to(DEV))
    assert torch.equal(packed.cpu(), G["pack.out"])                      # flux/__init__.py:25-31
    back = ops.flux_unpack(G["pack.out"].to(DEV), 16, 12, 20)
    assert torch.equal(back.cpu(), G["unpack.out"])                      # flux/__init__.py:34-45


def test_rope_kernel_vs_reference():
    """qk_norm_rope_fwd with the RMSNorm disabled reproduces _apply_rotary_emb_anyshape (flux/transformer.py:73-98).
    The reference rounds fp32(x*cos + rot*sin) to bf16 once; so does the kernel: identical up to fp32 FMA contraction,
    i.e. at most 1 bf16 ulp on isolated elements."""
    from simpletuner_amd import ops

    x = G["rope.x"].to(DEV)                      # [B, H, S, d]
    B, H, S, d = x.shape
    Sp = (S + 63) // 64 * 64
    D = H * d
    qkv = torch.zeros(B * S, 3 * D, device=DEV, dtype=torch.bfloat16)
    rows = x.permute(0, 2, 1, 3).reshape(B * S, D)
    qkv[:, :D] = rows; qkv[:, D:2 * D] = rows; qkv[:, 2 * D:] = rows
    Q = torch.zeros(B, H, S, d, device=DEV, dtype=torch.bfloat16); K = torch.zeros_like(Q)
    Qt = torch.zeros(B, H, d, Sp, device=DEV, dtype=torch.bfloat16); Kt = torch.zeros_like(Qt); Vt = torch.zeros_like(Qt)
    ops.qk_norm_rope_fwd(qkv, None, None, G["rope.cos"].to(DEV), G["rope.sin"].to(DEV), Q, K, Qt, Kt, Vt, B, H, d, S, 0, S, Sp)
    ref = G["rope.out_bf16"]
    diff = (Q.cpu().float() - ref.float()).abs()
    ulp = ref.float().abs().clamp_min(1e-3) * 2.0 ** -7
    frac_exact = (diff == 0).float().mean().item()
    print(f"[parity] rope vs reference: exact fraction={frac_exact:.5f}, max diff/ulp={(diff / ulp).max().item():.2f}")
    assert (diff <= ulp).all() and frac_exact > 0.99
    assert torch.equal(K, Q) and torch.equal(Vt[..., :S].cpu(), x.cpu().transpose(2, 3))


def test_flow_noise_mix_vs_reference_known_answers():
    from simpletuner_amd import ops

    x = G["flow.x"].to(torch.bfloat16); n = G["flow.n"].to(torch.bfloat16)
    xt, tg, _ = ops.flow_noise_mix(x.to(DEV), torch.full((2,), 0.25, device=DEV), noise=n.to(DEV))
    ref_xt = 0.75 * x.float() + 0.25 * n.float()          # tests/test_flux_model.py:122 (bf16 inputs)
    ref_tg = n.float() - x.float()                        # tests/test_flux_model.py:124
    assert torch.equal(xt.cpu(), ref_xt.to(torch.bfloat16)) and torch.equal(tg.cpu(), ref_tg.to(torch.bfloat16))


def test_ema_update_vs_reference_foreach():
    from simpletuner_amd import ops

    s = G["ema.s0"].to(DEV).clone(); p = G["ema.p"].to(DEV)
    ops.ema_update(s, p, 0.999)
    assert torch.allclose(s.cpu(), G["ema.s1_decay0.999"], atol=1e-6, rtol=0)    # ema.py:423; tolerance of tests/test_ema.py
