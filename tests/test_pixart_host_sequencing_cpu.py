"""The PixArt-Sigma ControlNet-Transformer engine's HOST SEQUENCING on the CPU (see tests/ops_emulator.py): trunk forward, and the trained ControlNet branch's
hand-written backward (copied blocks with 72 -> 80 padded heads, zero-init projections, `scale_shift_table` rows) against autograd on the oracle — the
configuration of BASELINE.json configs[4] at toy width.  The kernels are proven by tests/test_pixart_model_gpu.py; this runs the same engine code without a GPU."""
import pytest
import torch

from oracle.pixart import PixArtConfig, controlnet_forward, pixart_forward
from tests import ops_emulator as EMU

BF16 = torch.bfloat16
ARCH = dict(num_attention_heads=8, attention_head_dim=72, num_layers=4, caption_channels=128, sample_size=128, cross_attention_dim=576)


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _inputs(B=2, hw=(16, 16), Sk=20):
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(B, 4, *hw, generator=g).to(BF16)
    cond = torch.randn(B, 4, *hw, generator=g).to(BF16)
    enc = torch.randn(B, Sk, 128, generator=g).to(BF16)
    mask = torch.zeros(B, Sk); mask[0, :12] = 1; mask[1, :17] = 1
    t = torch.tensor([37.0, 820.0][:B])
    return lat, cond, enc, mask, t


def test_trunk_forward_through_the_emulator_matches_the_oracle(monkeypatch):
    EMU.install(monkeypatch)
    from simpletuner_amd.pixart.transformer import PixArtTransformer2DModel
    m = PixArtTransformer2DModel(device="cpu", **ARCH)
    m.init_synthetic(3)
    P = {k: v.detach().float() for k, v in m.named_parameters()}
    lat, cond, enc, mask, t = _inputs(hw=(16, 24))
    out = m(lat, encoder_hidden_states=enc, timestep=t, encoder_attention_mask=mask, return_dict=False)[0]
    ref = pixart_forward(P, PixArtConfig(**ARCH), lat.float(), enc.float(), mask, t, torch.tensor([[16.0, 24.0]]).expand(2, -1), torch.tensor([[16.0 / 24.0]]).expand(2, -1))
    assert out.shape == ref.shape == (2, 8, 16, 24) and _rel(out, ref) < 2e-2


def test_controlnet_branch_gradients_through_the_emulator_match_autograd(monkeypatch):
    EMU.install(monkeypatch)
    from simpletuner_amd.pixart.transformer import PixArtSigmaControlNetTransformerModel, PixArtTransformer2DModel
    m = PixArtTransformer2DModel(device="cpu", **ARCH)
    m.init_synthetic(5)
    cn = PixArtSigmaControlNetTransformerModel(m, num_layers=2)
    cn.init_adapter_synthetic(seed=9, std=0.05)
    with torch.no_grad():
        for blk, _ in cn.cblocks:
            for v in blk.P.values():
                v.add_(0.01 * torch.randn(v.shape, generator=torch.Generator().manual_seed(1)).to(BF16))
    P = {k: v.detach().float() for k, v in m.named_parameters()}
    C = {k: v.float().clone().requires_grad_(True) for k, v in cn.adapter_state_dict().items()}
    lat, cond, enc, mask, t = _inputs()
    target = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(4))
    EMU.BLOCK_CALLS.clear()
    out = cn(lat, encoder_hidden_states=enc, timestep=t, controlnet_cond=cond, encoder_attention_mask=mask, return_dict=False)[0]
    loss = ((out.chunk(2, dim=1)[0].float() - target) ** 2).mean()
    loss.backward()
    ref = controlnet_forward(P, C, PixArtConfig(**ARCH), 2, lat.float(), cond.float(), enc.float(), mask, t, torch.tensor([[16.0, 16.0]]).expand(2, -1), torch.tensor([[1.0]]).expand(2, -1))
    lref = ((ref.chunk(2, dim=1)[0] - target) ** 2).mean()
    assert _rel(out.detach(), ref.detach()) < 2e-2 and abs(loss.item() - lref.item()) < 2e-3 * max(1.0, abs(lref.item()))
    lref.backward()
    # every block ran through the block-level entry points (their emulation restates csrc/blocks.hip): st355_block_pixart_fwd / _bwd are the default path
    assert EMU.BLOCK_CALLS.get("pixart_fwd", 0) >= ARCH["num_layers"] + 2 and EMU.BLOCK_CALLS.get("pixart_bwd", 0) >= 2, EMU.BLOCK_CALLS
    names = {}
    for i, (blk, ex) in enumerate(cn.cblocks):
        for k, g in blk.G.items():
            names[f"controlnet_blocks.{i}.transformer_block.{k}"] = g
        for k, g in ex.G.items():
            names[f"controlnet_blocks.{i}.{k}"] = g
    assert set(names) == set(C)
    worst = (0.0, "")
    for k, g in names.items():
        if k.endswith("to_k.bias"):      # softmax-invariant: the true gradient is zero, both sides hold rounding noise
            continue
        r = _rel(g, C[k].grad)
        tol = 8e-2 if (k.endswith(".bias") or k.endswith("scale_shift_table")) else 6e-2
        worst = max(worst, (r / tol, f"{k}: {r:.3e}"))
        assert r < tol, (k, r)
    print(f"[emu] pixart controlnet host sequencing: {len(names)} tensors, worst (relative to its tolerance) {worst[1]}")


def test_controlnet_checkpoint_plans_through_the_emulator_are_bit_identical(monkeypatch):
    """the planner of pixart/transformer.py:627-700 over the ControlNet wrapper's loop units: recomputed segments (per layer, and interval 2 / stride 3) give the bit-identical
    prediction and adapter gradient arena"""
    EMU.install(monkeypatch)
    from simpletuner_amd.pixart.transformer import PixArtSigmaControlNetTransformerModel, PixArtTransformer2DModel

    def run(ckpt, interval=None, stride=None):
        m = PixArtTransformer2DModel(device="cpu", **ARCH)
        m.init_synthetic(5)
        cn = PixArtSigmaControlNetTransformerModel(m, num_layers=3)
        cn.init_adapter_synthetic(seed=9, std=0.05)
        if ckpt:
            cn.enable_gradient_checkpointing()
            cn.set_gradient_checkpointing_interval(interval)
            cn.set_gradient_checkpointing_segment_stride(stride)
        lat, cond, enc, mask, t = _inputs()
        out = cn(lat, encoder_hidden_states=enc, timestep=t, controlnet_cond=cond, encoder_attention_mask=mask, return_dict=False)[0]
        (out.float() ** 2).mean().backward()
        return out.detach().clone(), cn.grad_arena.detach().clone()

    o0, g0 = run(False)
    for plan in ((None, None), (2, 3)):
        o1, g1 = run(True, *plan)
        assert torch.equal(o0, o1) and torch.equal(g0, g1) and g0.float().abs().sum().item() > 0


def test_trunk_forward_with_tokenwise_timesteps_through_the_emulator_matches_the_oracle(monkeypatch):
    """TOKENWISE timesteps [B, S] (CREPA self-flow; reference tests/test_pixart_model.py:91-115; oracle branch pinned to the executed reference class): per-token
    AdaLN-single rows in every block and in the head (rows_per_batch = 1), the size conditions shared by a sample's tokens"""
    EMU.install(monkeypatch)
    from simpletuner_amd.pixart.transformer import PixArtTransformer2DModel
    m = PixArtTransformer2DModel(device="cpu", **ARCH)
    m.init_synthetic(3)
    P = {k: v.detach().float() for k, v in m.named_parameters()}
    lat, cond, enc, mask, t = _inputs()
    B, S = lat.shape[0], (lat.shape[2] // 2) * (lat.shape[3] // 2)
    tt = torch.rand(B, S, generator=torch.Generator().manual_seed(8)) * 900.0 + 50.0
    EMU.BLOCK_CALLS.clear()
    out = m(lat, encoder_hidden_states=enc, timestep=tt, encoder_attention_mask=mask, return_dict=False)[0]
    assert EMU.BLOCK_CALLS.get("pixart_fwd", 0) == 0            # per-token rows: the host-side sequencing, not the per-sample C entry point
    res = torch.tensor([[float(lat.shape[2]), float(lat.shape[3])]]).expand(B, -1)
    ar = torch.tensor([[lat.shape[2] / lat.shape[3]]]).expand(B, -1)
    ref = pixart_forward(P, PixArtConfig(**ARCH), lat.float(), enc.float(), mask, tt, res, ar)
    flat = pixart_forward(P, PixArtConfig(**ARCH), lat.float(), enc.float(), mask, tt.mean(dim=1), res, ar)
    assert _rel(out, ref) < 2e-2 and _rel(flat, ref) > 5e-2, (_rel(out, ref), _rel(flat, ref))
    with pytest.raises(ValueError, match="tokenwise timestep embedding expected shape"):
        m(lat, encoder_hidden_states=enc, timestep=tt[:, :5], encoder_attention_mask=mask, return_dict=False)


def _true_lora(model):
    """the adapters of the HIP model in their true (peft) shapes, as float leaves for the oracle: {module: (A [r, in], B [out, r])}"""
    sd = model.lora_state_dict()
    out = {}
    for k, v in sd.items():
        mod, which = k.split(".lora_")
        out.setdefault(mod, [None, None])[0 if which.startswith("A") else 1] = v.float().clone().requires_grad_(True)
    return {k: tuple(v) for k, v in out.items()}


@pytest.mark.parametrize("route", [False, True])
def test_trunk_lora_gradients_through_the_emulator_match_autograd(monkeypatch, route):
    """PixArt LoRA (pixart/model.py:59: to_k, to_q, to_v, to_out.0 of attn1 and attn2 in every block) on the trunk: adapters in the K-extension of the head-padded
    projections, prediction and every adapter gradient — compared in their true (un-padded) shapes — against autograd on the oracle; the pad rows / columns of the
    working-layout factors keep zero gradients.  route: TREAD on the trunk (pixart/transformer.py:487-489), half of the tokens routed around blocks [1, -2], the
    oracle replaying the same permutation."""
    EMU.install(monkeypatch)
    from simpletuner_amd.pixart.transformer import HP, PixArtTransformer2DModel
    from simpletuner_amd.training.tread import ReplayRouter
    m = PixArtTransformer2DModel(device="cpu", **ARCH)
    m.init_synthetic(5)
    m.add_lora_adapter(rank=8, alpha=16.0, init_b_std=0.05)
    lat, cond, enc, mask, t = _inputs()
    B, S = lat.shape[0], (lat.shape[2] // 2) * (lat.shape[3] // 2)
    routes, rec = [], None
    if route:
        g = torch.Generator().manual_seed(11)
        perm = torch.stack([torch.randperm(S, generator=g) for _ in range(B)])
        K = S - int(round(S * 0.5))
        rec = {"mask": torch.ones(B, S, dtype=torch.bool).scatter_(1, perm[:, :K], False), "ids_keep": perm[:, :K], "ids_mask": perm[:, K:], "ids_shuffle": perm,
               "ids_restore": torch.argsort(perm, dim=1)}
        routes = [{"selection_ratio": 0.5, "start_layer_idx": 1, "end_layer_idx": -2}]
        m.set_router(ReplayRouter([rec]), routes)
    m.train()
    target = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(4))
    EMU.BLOCK_CALLS.clear()
    out = m(lat, encoder_hidden_states=enc, timestep=t, encoder_attention_mask=mask, return_dict=False)[0]
    assert EMU.BLOCK_CALLS.get("pixart_fwd", 0) == 0               # adapters ride in the K-extension: the host-side sequencing, not the adapter-less C entry point
    loss = ((out.chunk(2, dim=1)[0].float() - target) ** 2).mean()
    loss.backward()
    P = {k: v.detach().float() for k, v in m.named_parameters() if ".lora_" not in k}
    lp = _true_lora(m)
    assert len(lp) == ARCH["num_layers"] * 8
    res, ar = torch.tensor([[16.0, 16.0]]).expand(2, -1), torch.tensor([[1.0]]).expand(2, -1)
    ref = pixart_forward(P, PixArtConfig(**ARCH), lat.float(), enc.float(), mask, t, res, ar, lora=lp, lora_scale=16.0 / 8,
                         tread={"routes": routes, "mask_infos": [rec]} if route else None)
    lref = ((ref.chunk(2, dim=1)[0] - target) ** 2).mean()
    lref.backward()
    assert _rel(out.detach(), ref.detach()) < 2e-2 and abs(loss.item() - lref.item()) < 2e-3 * max(1.0, abs(lref.item()))
    H, hd = ARCH["num_attention_heads"], ARCH["attention_head_dim"]
    worst = (0.0, "")
    for name, p in m.named_parameters():
        if ".lora_" not in name:
            continue
        mod, which = name.split(".lora_")
        g = p.grad
        if which.startswith("A") and g.shape[1] == H * HP:          # head-padded input axis: the pad columns carry no gradient
            g3 = g.view(g.shape[0], H, HP)
            assert float(g3[:, :, hd:].abs().max()) == 0
            g = g3[:, :, :hd].reshape(g.shape[0], H * hd)
        if which.startswith("B") and g.shape[0] == H * HP:
            g3 = g.view(H, HP, g.shape[1])
            assert float(g3[:, hd:].abs().max()) == 0
            g = g3[:, :hd].reshape(H * hd, g.shape[1])
        want = lp[mod][0 if which.startswith("A") else 1].grad
        r = _rel(g, want)
        worst = max(worst, (r, name))
        assert r < 6e-2, (name, r)
    print(f"[emu] pixart trunk LoRA{' + TREAD' if route else ''}: pred rel_l2={_rel(out.detach(), ref.detach()):.3e}, worst adapter gradient rel_l2={worst[0]:.3e} at {worst[1]}")


def test_pixart_lora_file_round_trip_in_true_peft_shapes(tmp_path):
    """save_lora_weights writes the adapters in their TRUE shapes ([r, 1152] / [1152, r], diffusers `transformer.` prefix: what the reference's pipelines load); loading
    them back fills the head-padded working layout with zero pad lanes"""
    from types import SimpleNamespace

    from safetensors.torch import load_file

    from simpletuner_amd.pixart.model import PixartSigma
    from simpletuner_amd.pixart.transformer import HP, PixArtTransformer2DModel
    m = PixArtTransformer2DModel(device="cpu", **ARCH)
    m.add_lora_adapter(rank=4, alpha=4.0, init_b_std=0.1)
    plug = PixartSigma.__new__(PixartSigma)
    plug.config, plug.accelerator, plug.model, plug.controlnet = SimpleNamespace(lora_rank=4, lora_alpha=4.0, lora_format=None), SimpleNamespace(device=torch.device("cpu")), m, None
    plug.save_lora_weights(str(tmp_path))
    flat = load_file(str(tmp_path / plug.LORA_WEIGHT_NAME))
    H, hd = ARCH["num_attention_heads"], ARCH["attention_head_dim"]
    assert len(flat) == ARCH["num_layers"] * 8 * 2
    assert flat["transformer.transformer_blocks.0.attn1.to_q.lora_B.weight"].shape == (H * hd, 4)
    assert flat["transformer.transformer_blocks.1.attn2.to_out.0.lora_A.weight"].shape == (4, H * hd)
    before = {n: p.detach().clone() for n, p in m.named_parameters() if ".lora_" in n}
    with torch.no_grad():
        for n, p in m.named_parameters():
            if ".lora_" in n:
                p.fill_(7.0)
    plug.load_lora_weights(input_dir=str(tmp_path))
    for n, p in m.named_parameters():
        if ".lora_" in n:
            assert torch.equal(p, before[n]), n
    b = dict(m.named_parameters())["transformer_blocks.0.attn1.to_k.lora_B.default.weight"]
    assert b.shape == (H * HP, 4) and float(b.detach().view(H, HP, 4)[:, hd:].abs().max()) == 0
