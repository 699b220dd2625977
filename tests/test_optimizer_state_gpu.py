"""Checkpoint / resume continuity on the MI355X (optimizer state, EMA shadow): N steps straight through equal, BIT FOR BIT,
K steps + state_dict -> fresh objects -> load_state_dict + (N-K) steps, for the fused AdamW, the fused AdamWBF16 (stochastic rounding is
counter-based: the draw of step s depends only on (seed, s), so it resumes exactly) and the EMA shadow (`ema_model.pt` file)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    arena = (torch.randn(4096 + 1024, generator=g) * 0.05).to(BF16).to(dev)
    grads = torch.zeros_like(arena)
    ps = [torch.nn.Parameter(arena[:4096].view(64, 64)), torch.nn.Parameter(arena[4096:].view(1024))]
    off = 0
    for p in ps:
        p.grad = grads[off:off + p.numel()].view_as(p)
        off += p.numel()
    return arena, grads, ps


def _set_grads(grads, step):
    g = torch.Generator().manual_seed(1000 + step)
    grads.copy_((torch.randn(grads.numel(), generator=g) * 1e-2).to(BF16))


@pytest.mark.parametrize("which", ["adamw", "adamw_bf16"])
def test_optimizer_resume_is_bit_exact(which):
    from simpletuner_amd.training.optimizer import St355AdamW, St355AdamWBF16
    dev = torch.device("cuda", 0)

    def make(ps):
        if which == "adamw":
            return St355AdamW(ps, lr=1e-3, weight_decay=0.01)
        torch.manual_seed(0)                                   # the random initial decay phase of every tensor (reference :80-84)
        return St355AdamWBF16(ps, lr=1e-3, weight_decay=0.5, seed=7)      # large decay: the delayed decay fires inside 4 steps

    a_arena, a_grads, a_ps = _params(dev, 1)
    a = make(a_ps)
    for s in range(1, 5):
        _set_grads(a_grads, s)
        a.step()
    b_arena, b_grads, b_ps = _params(dev, 1)
    b = make(b_ps)
    for s in range(1, 3):
        _set_grads(b_grads, s)
        b.step()
    sd = b.state_dict()
    sd = {"state": {k: {n: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for n, v in st.items()} for k, st in sd["state"].items()},
          "param_groups": sd["param_groups"]}                  # as read back from a checkpoint file
    c_arena, c_grads, c_ps = _params(dev, 99)                  # fresh objects, different init ...
    c_arena.copy_(b_arena)                                     # ... the model weights come from the checkpoint
    c = make(c_ps)
    c.load_state_dict(sd)
    for s in range(3, 5):
        _set_grads(c_grads, s)
        c.step()
    assert torch.equal(c_arena, a_arena)
    fa, fc = a._flat[0], c._flat[0]
    assert fc["step"] == fa["step"] == 4 and torch.equal(fc["m"], fa["m"]) and torch.equal(fc["v"], fa["v"])
    if which == "adamw_bf16":
        assert torch.equal(fc["shift"], fa["shift"])
        assert [c.state[p]["accumulated_decay"] for p in c_ps] == pytest.approx([a.state[p]["accumulated_decay"] for p in a_ps], abs=1e-12)


def test_ema_resume_from_file_is_bit_exact(tmp_path):
    from types import SimpleNamespace

    from simpletuner_amd.training.ema import EMAModel
    dev = torch.device("cuda", 0)
    acc = SimpleNamespace(process_index=0)

    def drift(arena, s):
        g = torch.Generator().manual_seed(50 + s)
        arena.add_((torch.randn(arena.numel(), generator=g) * 1e-2).to(BF16).to(dev))

    a_arena, _, a_ps = _params(dev, 2)
    ea = EMAModel(SimpleNamespace(), acc, a_ps, decay=0.99)
    for s in range(1, 7):
        drift(a_arena, s)
        ea.step(a_ps, global_step=s)
    b_arena, _, b_ps = _params(dev, 2)
    eb = EMAModel(SimpleNamespace(), acc, b_ps, decay=0.99)
    for s in range(1, 4):
        drift(b_arena, s)
        eb.step(b_ps, global_step=s)
    eb.save_state_dict(str(tmp_path / "ema" / "ema_model.pt"))
    c_arena, _, c_ps = _params(dev, 77)
    c_arena.copy_(b_arena)
    ec = EMAModel(SimpleNamespace(), acc, c_ps, decay=0.5)
    ec.load_state_dict(str(tmp_path / "ema" / "ema_model.pt"))
    assert ec.decay == 0.99 and ec.optimization_step == 3
    for s in range(4, 7):
        drift(c_arena, s)
        ec.step(c_ps, global_step=s)
    assert torch.equal(ec.shadow_flat, ea.shadow_flat) and ec.optimization_step == 6
