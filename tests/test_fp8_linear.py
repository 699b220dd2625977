"""fp8-native Linear (K19; helpers/training/quantisation/fp8_native.py:25-119): oracle pinned to the reference module executed in this
container (tests/golden/fp8_vectors.pt: weight quantisation from the reference function; activation bytes and the scale vectors as the
reference's own forward hands them to torch._scaled_mm), HIP kernels pinned to the same vectors; the fp8 GEMM additionally checked
against torch._scaled_mm on the MI355X itself when the installed torch supports the row-wise mode there."""
from pathlib import Path

import pytest
import torch

from oracle import train_math as TM

G = torch.load(Path(__file__).parent / "golden" / "fp8_vectors.pt")


def test_oracle_quantisers_match_reference_bitwise():
    q, sc = TM.fp8_quantize_weight(G["w"])
    assert torch.equal(q.view(torch.uint8), G["w_q"]) and torch.equal(sc, G["w_scale"])
    x2 = G["x"].reshape(-1, G["x"].shape[-1])
    xq, sa = TM.fp8_quantize_act(x2)
    assert torch.equal(xq.view(torch.uint8), G["x_q"])
    assert torch.equal(sa.expand(x2.shape[0]).reshape(-1, 1), G["scale_a"])
    assert torch.equal(G["scale_b"].reshape(-1), G["w_scale"])
    out = TM.fp8_linear(xq, sa, q, sc, G["bias"]).reshape(G["out_fp32_semantics"].shape)
    assert torch.equal(out, G["out_fp32_semantics"])


@pytest.mark.gpu
def test_hip_fp8_quantisers_and_linear():
    from simpletuner_amd import ops
    dev = "cuda:0"
    q, sc = ops.fp8_quantize_weight(G["w"].to(dev))
    assert torch.equal(q.cpu(), G["w_q"]), f"{(q.cpu() != G['w_q']).sum().item()} weight bytes differ"
    assert torch.equal(sc.cpu(), G["w_scale"])
    x2 = G["x"].reshape(-1, G["x"].shape[-1]).to(dev).contiguous()
    xq, sa = ops.fp8_quantize_act(x2)
    assert torch.equal(xq.cpu(), G["x_q"]), f"{(xq.cpu() != G['x_q']).sum().item()} activation bytes differ"
    assert sa.item() == G["scale_a"][0].item()
    # the contraction: K multiple of 128 is required by the kernel -> 384 ok; N = 264
    out = ops.linear_fp8(xq, sa, q, sc, bias=G["bias"].to(dev))
    ref = G["out_fp32_semantics"].reshape(-1, 264).float()
    d = (out.float().cpu() - ref).abs()
    assert d.max().item() <= 2.0 ** -7 * ref.abs().max().item(), d.max().item()        # same fp8 operands, fp32 accumulate: <= 1 bf16 ulp of the largest output
    assert (out.cpu() == G["out_fp32_semantics"].reshape(-1, 264)).float().mean().item() > 0.98


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(4608, 3072, 3072), (1000, 520, 256), (18432, 12288, 3072)])
def test_hip_fp8_linear_large_vs_oracle_and_torch_scaled_mm(M, N, K):
    from simpletuner_amd import ops
    dev = "cuda:0"
    torch.manual_seed(9)
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    q, sc = ops.fp8_quantize_weight(w)
    xq, sa = ops.fp8_quantize_act(x)
    out = ops.linear_fp8(xq, sa, q, sc, bias=bias)
    rows = slice(0, min(M, 512))
    ref = TM.fp8_linear(xq[rows].cpu().view(torch.float8_e5m2), sa.cpu(), q.cpu().view(torch.float8_e4m3fn), sc.cpu(), bias.cpu())
    rel = ((out[rows].float().cpu() - ref.float()).norm() / ref.float().norm()).item()
    assert rel < 4e-3, rel
    # and against the un-quantised product: the fp8 error itself (e5m2 inputs: 2 mantissa bits)
    full = (x[rows].float() @ w.float().t() + bias.float()).cpu()
    assert ((out[rows].float().cpu() - full).norm() / full.norm()).item() < 8e-2
    try:    # what the reference executes on a GPU: the same operands through torch._scaled_mm (row-wise scales)
        t = torch._scaled_mm(xq.view(torch.float8_e5m2), q.view(torch.float8_e4m3fn).t(), scale_a=sa.expand(M).reshape(M, 1).contiguous(),
                             scale_b=sc.reshape(1, N).contiguous(), bias=bias, out_dtype=torch.bfloat16, use_fast_accum=True)
    except Exception as e:       # noqa: BLE001 - capability probe
        pytest.skip(f"torch._scaled_mm row-wise mode unavailable on this stack: {str(e)[:120]}")
    relt = ((out.float() - t.float()).norm() / t.float().norm()).item()
    print(f"[fp8] {M}x{N}x{K}: vs torch._scaled_mm rel {relt:.2e}")
    assert relt < 4e-3
