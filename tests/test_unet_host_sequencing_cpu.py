"""The UNet engine's HOST SEQUENCING on the CPU (see tests/ops_emulator.py): the SDXL / SD 1.5 `UNet2DConditionModel` of BASELINE.json configs[0] / [1] — convolution-as-GEMM over
zero-bordered grid buffers, GroupNorm, GEGLU, cross-attention (fused, zero-padded and unfused head widths), the tape-driven backward — against autograd on the oracle, at toy widths.
The kernels are proven by tests/test_unet_kernels_gpu.py / test_unet_model_gpu.py; this runs the same engine code without a GPU."""
import pytest
import torch

from oracle.unet import UNetConfig, unet_forward
from tests import ops_emulator as EMU

BF16 = torch.bfloat16
SMALL = dict(block_out_channels=(64, 128), layers_per_block=1, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
             up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2), attention_head_dim=(1, 2), cross_attention_dim=128,
             projection_class_embeddings_input_dim=64 + 6 * 64, addition_time_embed_dim=64)
SD15_NARROW = dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                   up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D"), transformer_layers_per_block=(1, 1), attention_head_dim=(8, 8),
                   cross_attention_dim=128, use_linear_projection=False, addition_embed_type=None)
SD15_WIDE = dict(block_out_channels=(64, 320), layers_per_block=1, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
                 up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 1), attention_head_dim=(1, 2),
                 cross_attention_dim=128, use_linear_projection=False, addition_embed_type=None)
ARCHS = {"sdxl_small": SMALL, "sd15_narrow_heads": SD15_NARROW, "sd15_wide_heads": SD15_WIDE}


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _inputs(B, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(B, 4, H, W, generator=g).to(BF16)
    t = torch.tensor([17.0, 801.0, 333.0, 950.0][:B])
    ehs = torch.randn(B, 9, 128, generator=g).to(BF16)
    te = torch.randn(B, 64, generator=g).to(BF16)
    ti = torch.tensor([[64.0, 48.0, 0.0, 0.0, 64.0, 48.0]] * B).to(BF16)
    return sample, t, ehs, te, ti


def _unet(monkeypatch, arch, seed):
    EMU.install(monkeypatch)
    from simpletuner_amd.unet.unet import UNet2DConditionModel
    m = UNet2DConditionModel(device="cpu", **ARCHS[arch])
    m.init_synthetic(seed)
    return m, UNet2DConditionModel


def test_forward_through_the_emulator_matches_the_oracle(monkeypatch):
    m, _ = _unet(monkeypatch, "sdxl_small", 3)
    P = {k: v.float() for k, v in m.diffusers_state_dict().items()}
    sample, t, ehs, te, ti = _inputs(2, 16, 24)
    out = m(sample, t, ehs, None, added_cond_kwargs={"text_embeds": te, "time_ids": ti}, return_dict=False)[0]
    ref = unet_forward(P, UNetConfig(**SMALL), sample.float(), t, ehs.float(), {"text_embeds": te.float(), "time_ids": ti.float()})
    assert out.shape == ref.shape and _rel(out, ref) < 2e-2


@pytest.mark.parametrize("arch", list(ARCHS))
def test_full_finetune_gradients_through_the_emulator_match_autograd(monkeypatch, arch):
    m, Cls = _unet(monkeypatch, arch, 5)
    m.enable_full_finetune()
    P = {k: v.float().clone().requires_grad_(True) for k, v in m.diffusers_state_dict().items()}
    sample, t, ehs, te, ti = _inputs(2, 16, 16, seed=1)
    target = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(9))
    ack = {"text_embeds": te, "time_ids": ti} if arch == "sdxl_small" else None
    out = m(sample, t, ehs, None, added_cond_kwargs=ack, return_dict=False)[0]
    loss = ((out.float() - target) ** 2).mean()
    loss.backward()
    ref = unet_forward(P, UNetConfig(**ARCHS[arch]), sample.float(), t, ehs.float(), {"text_embeds": te.float(), "time_ids": ti.float()})
    assert _rel(out.detach(), ref.detach()) < 2e-2
    lref = ((ref - target) ** 2).mean()
    lref.backward()
    assert abs(loss.item() - lref.item()) < 2e-3 * max(1.0, abs(lref.item()))
    g = Cls(device="cpu", **ARCHS[arch])
    g.load_diffusers_state({k: v.grad for k, v in P.items()})
    worst = (0.0, "")
    for s, sg in zip(m._specs, g._specs):
        got, want = s.g.float(), sg.t.float()
        if s.name.startswith("conv_in.weight"):
            got = got[:, :72].reshape(-1, 9, 8)[:, :, :4]; want = want[:, :72].reshape(-1, 9, 8)[:, :, :4]
        if s.name.startswith("conv_out"):
            got, want = got[:4], want[:4]
        r = _rel(got, want)
        tol = 8e-2 if s.kind != "w" else 6e-2
        worst = max(worst, (r / tol, f"{s.name}: {r:.3e}"))
        assert r < tol, (s.name, r)
    print(f"[emu] unet {arch} host sequencing: {len(m._specs)} tensors, worst (relative to its tolerance) {worst[1]}")


def test_lora_gradients_through_the_emulator_match_autograd(monkeypatch):
    rank, alpha = 16, 16.0
    m, _ = _unet(monkeypatch, "sdxl_small", 6)
    m.add_lora_adapter(rank=rank, alpha=alpha, seed=3, init_b_std=0.05)
    P = {k: v.float() for k, v in m.diffusers_state_dict().items()}
    lora = {n: p.detach().float().clone().requires_grad_(True) for n, p in m.named_parameters() if ".lora_" in n}
    Pe = dict(P)
    for n in lora:
        if ".lora_A." in n:
            base = n.replace(".lora_A.default.weight", "")
            Pe[base + ".weight"] = P[base + ".weight"] + (alpha / rank) * lora[base + ".lora_B.default.weight"] @ lora[n]
    sample, t, ehs, te, ti = _inputs(2, 16, 16, seed=2)
    target = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(8))
    out = m(sample, t, ehs, None, added_cond_kwargs={"text_embeds": te, "time_ids": ti}, return_dict=False)[0]
    loss = ((out.float() - target) ** 2).mean()
    loss.backward()
    ref = unet_forward(Pe, UNetConfig(**SMALL), sample.float(), t, ehs.float(), {"text_embeds": te.float(), "time_ids": ti.float()})
    assert _rel(out.detach(), ref.detach()) < 2e-2
    ((ref - target) ** 2).mean().backward()
    for n, p in m.named_parameters():
        if ".lora_" in n:
            assert _rel(p.grad, lora[n].grad) < 6e-2, (n, _rel(p.grad, lora[n].grad))


@pytest.mark.parametrize("interval,stride,lora", [(None, None, False), (2, 3, True)])
def test_checkpoint_plans_through_the_emulator_are_bit_identical(monkeypatch, interval, stride, lora):
    """diffusers' `enable_gradient_checkpointing` over the UNet's units (every ResnetBlock2D / transformer its own checkpoint, or the interval / stride plans): a checkpointed
    unit runs without a tape and is re-run on a private tape in backward — prediction and the gradient arena are bit-identical to the run that records everything"""
    def run(ckpt):
        m, _ = _unet(monkeypatch, "sdxl_small", 5)
        if lora:
            m.add_lora_adapter(rank=8, alpha=8.0, seed=4, init_b_std=0.05)
        else:
            m.enable_full_finetune()
        if ckpt:
            m.enable_gradient_checkpointing()
            m.set_gradient_checkpointing_interval(interval)
            m.set_gradient_checkpointing_segment_stride(stride)
        sample, t, ehs, te, ti = _inputs(2, 16, 16, seed=1)
        target = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(9))
        out = m(sample, t, ehs, None, added_cond_kwargs={"text_embeds": te, "time_ids": ti}, return_dict=False)[0]
        ((out.float() - target) ** 2).mean().backward()
        return out.detach().clone(), torch.cat([p.grad.detach().reshape(-1).float() for p in m.trainable_parameters()]).clone()

    o0, g0 = run(False)
    o1, g1 = run(True)
    assert torch.equal(o0, o1) and torch.equal(g0, g1) and g0.abs().sum().item() > 0


def test_head_widths_pick_the_padded_kernel_width_the_contract_allows():
    """include/st355.h: head_dim 96 is the width of a zero-padded narrower head whose channels [80, 96) are zero — the 64-row kernels contract over 80 channels.  The UNet's
    attention therefore pads heads of up to 80 channels to 96 and anything wider (81 … 128) to 128; 64 and below to 64; above 128 the unfused path."""
    import inspect

    from simpletuner_amd.unet import unet as U
    src = inspect.getsource(U.UNet2DConditionModel._attention)
    ns = {}
    line = next(l for l in src.splitlines() if l.strip().startswith("hp = "))
    for hd, want in ((40, 64), (64, 64), (72, 96), (80, 96), (81, 128), (96, 128), (128, 128), (160, 0)):
        exec(line.strip().split("#")[0], {"hd": hd}, ns)
        assert ns["hp"] == want, (hd, ns["hp"], want)
