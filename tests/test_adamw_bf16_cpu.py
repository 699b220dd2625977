"""AdamWBF16 (the examples' default optimizer) — the oracle restatement pinned BIT FOR BIT to the reference class executed in this
container (tools/gen_golden.py::gen_adamw_bf16 -> tests/golden/adamw_bf16_vectors.pt: 4 steps, two tensors, every stochastic-
rounding draw recorded)."""
from pathlib import Path

import pytest
import torch

from oracle import train_math as TM

G = torch.load(Path(__file__).parent / "golden" / "adamw_bf16_vectors.pt")


def test_stochastic_round_is_truncation_of_biased_bits():
    x = torch.tensor([1.0, 1.00390625, -2.5, 3.14159], dtype=torch.float32)
    assert torch.equal(TM.stochastic_round_bf16(x, torch.zeros(4, dtype=torch.int32)).float(), torch.tensor([1.0, 1.0, -2.5, 3.140625]))
    up = TM.stochastic_round_bf16(x, torch.full((4,), 65535, dtype=torch.int32)).float()
    assert up[1] == 1.0078125 and up[3] == 3.15625 and up[0] == 1.0       # exact bf16 values never move


def test_decay_schedule_matches_reference_state():
    acc = [r * G["decay_threshold"] for r in G["accumulated_decay0"]]
    for st in G["steps"]:
        for i in range(2):
            _, acc[i] = TM.adamw_bf16_decay_schedule(acc[i], G["wd"], G["lr"], G["decay_threshold"])
            assert abs(acc[i] - st["accumulated_decay"][i]) < 1e-9


def test_oracle_reproduces_reference_states_bitwise():
    n = len(G["p0"])
    p = [t.clone() for t in G["p0"]]
    m = [torch.zeros_like(t) for t in p]; v = [torch.zeros_like(t) for t in p]; sh = [torch.zeros_like(t) for t in p]
    acc = [r * G["decay_threshold"] for r in G["accumulated_decay0"]]
    b1, b2 = G["betas"]
    applied = 0
    for k, st in enumerate(G["steps"]):
        for i in range(n):
            dec, acc[i] = TM.adamw_bf16_decay_schedule(acc[i], G["wd"], G["lr"], G["decay_threshold"])
            applied += dec > 0
            p[i], m[i], v[i], sh[i] = TM.adamw_bf16_step(p[i], st["grads"][i], m[i], v[i], sh[i], k + 1, G["lr"], b1, b2, G["eps"], dec,
                                                         st["draws"][i], decay_alpha="aten_cpu")
            for name, mine, ref in (("p", p[i], st["p"][i]), ("exp_avg", m[i], st["exp_avg"][i]), ("exp_avg_sq", v[i], st["exp_avg_sq"][i]),
                                    ("shift", sh[i], st["shift"][i])):
                assert torch.equal(mine.view(torch.int16), ref.view(torch.int16)), f"step {k + 1} tensor {i} {name}: " \
                    f"{(mine.view(torch.int16) != ref.view(torch.int16)).sum().item()} of {ref.numel()} elements differ"
    assert applied >= 2          # the delayed weight decay fired inside the fixture


def test_decay_alpha_modes_differ_only_by_the_alpha_rounding():
    """the GPU-semantics mode (fp32 alpha; what the HIP kernel implements) vs the ATen-CPU quirk mode: same states except `shift` on the
    steps where the delayed decay fires, and there by no more than the bf16 rounding of alpha (2^-8 relative on the decay term) plus one
    rounding of the result."""
    st = G["steps"][0]
    p0 = G["p0"][0]; z = torch.zeros_like(p0)
    dec, _ = TM.adamw_bf16_decay_schedule(G["accumulated_decay0"][0] * G["decay_threshold"], G["wd"], G["lr"], G["decay_threshold"])
    assert dec > 0
    b1, b2 = G["betas"]
    a = TM.adamw_bf16_step(p0, st["grads"][0], z, z, z, 1, G["lr"], b1, b2, G["eps"], dec, st["draws"][0], decay_alpha="fp32")
    b = TM.adamw_bf16_step(p0, st["grads"][0], z, z, z, 1, G["lr"], b1, b2, G["eps"], dec, st["draws"][0], decay_alpha="aten_cpu")
    for k in range(3):
        assert torch.equal(a[k], b[k])
    diff = (a[3].float() - b[3].float()).abs()
    bound = dec * a[0].float().abs() * 2.0 ** -7 + a[3].float().abs() * 2.0 ** -7 + 1e-9
    assert (diff <= bound).all() and (diff > 0).any()


def test_optimizer_state_dict_resume_keeps_flat_arenas_and_fp32_moments():
    """accelerator.save_state / load_state round trip (torch.optim.Optimizer.state_dict format): the loaded moments land INSIDE the flat arenas the
    one-launch step reads, fp32 moments of bf16 parameters stay fp32 (torch's own loader would cast them to the parameter dtype), step counters and
    AdamWBF16's per-tensor owed decay continue.  (The step itself is a HIP launch: tests/test_optimizer_state_gpu.py.)"""
    import torch

    from simpletuner_amd.training.optimizer import St355AdamW, St355AdamWBF16
    arena = torch.zeros(24, dtype=torch.bfloat16)
    ps = [torch.nn.Parameter(arena[:8].view(2, 4)), torch.nn.Parameter(arena[8:].view(4, 4))]
    a = St355AdamW(ps, lr=3e-4, weight_decay=0.02)
    st = a._group_flat(0, a.param_groups[0])
    st["m"].copy_(torch.arange(24, dtype=torch.float32) * 1e-5 + 1.0 / 3.0)       # not representable in bf16
    st["v"].copy_(torch.arange(24, dtype=torch.float32) * 1e-7 + 1e-3)
    st["step"] = 5
    for p in ps:
        a.state[p]["step"] = torch.tensor(5.0)
    sd = a.state_dict()
    assert set(sd["state"][1]) == {"step", "exp_avg", "exp_avg_sq"} and sd["param_groups"][0]["lr"] == 3e-4
    arena2 = torch.zeros(24, dtype=torch.bfloat16)
    ps2 = [torch.nn.Parameter(arena2[:8].view(2, 4)), torch.nn.Parameter(arena2[8:].view(4, 4))]
    b = St355AdamW(ps2, lr=1.0)
    b.load_state_dict(sd)
    sb = b._flat[0]
    assert sb["ok"] and sb["step"] == 5 and sb["m"].dtype == torch.float32 and torch.equal(sb["m"], st["m"]) and torch.equal(sb["v"], st["v"])
    assert b.param_groups[0]["lr"] == 3e-4 and b.param_groups[0]["weight_decay"] == 0.02
    assert b.state[ps2[1]]["exp_avg"].data_ptr() == sb["m"][8:].data_ptr() and float(b.state[ps2[0]]["step"]) == 5.0
    with pytest.raises(ValueError, match="parameter group"):
        St355AdamW([torch.nn.Parameter(torch.zeros(3))]).load_state_dict(sd)

    c = St355AdamWBF16(ps, lr=1e-4, weight_decay=0.01)
    sc = c._init_group(0, c.param_groups[0])
    sc["m"].copy_(torch.arange(24).to(torch.bfloat16)); sc["v"].fill_(0.25); sc["shift"].fill_(-0.5); sc["step"] = 9
    for i, p in enumerate(ps):
        c.state[p]["step"] = 9.0
        c.state[p]["accumulated_decay"] = 1e-3 * (i + 1)
    sd2 = c.state_dict()
    assert set(sd2["state"][0]) == {"step", "exp_avg", "exp_avg_sq", "shift", "accumulated_decay"}
    d = St355AdamWBF16(ps2, lr=1.0)
    d.load_state_dict(sd2)
    sdd = d._flat[0]
    assert sdd["step"] == 9 and torch.equal(sdd["m"], sc["m"]) and torch.equal(sdd["v"], sc["v"]) and torch.equal(sdd["shift"], sc["shift"])
    assert [d.state[p]["accumulated_decay"] for p in ps2] == [1e-3, 2e-3] and d.state[ps2[0]]["step"] == 9.0
    assert d.state[ps2[1]]["shift"].data_ptr() == sdd["shift"][8:].data_ptr() and d.param_groups[0]["lr"] == 1e-4


def test_decay_phases_are_replica_identical_whatever_the_global_rng():
    """St355AdamWBF16 draws each tensor's initial delayed-decay phase from (optimizer seed, group), not from the rank-dependent global RNG"""
    from simpletuner_amd.training.optimizer import St355AdamWBF16

    def phases(global_seed, opt_seed):
        torch.manual_seed(global_seed)
        arena = torch.zeros(24, dtype=torch.bfloat16)
        ps = [torch.nn.Parameter(arena[:8].view(2, 4)), torch.nn.Parameter(arena[8:].view(4, 4))]
        o = St355AdamWBF16(ps, lr=1e-4, weight_decay=0.01, seed=opt_seed)
        o._init_group(0, o.param_groups[0])
        return [o.state[p]["accumulated_decay"] for p in ps]

    a, b, c = phases(42, 7), phases(43, 7), phases(42, 8)
    assert a == b and a != c and all(0.0 <= x < St355AdamWBF16.decay_threshold for x in a) and a[0] != a[1]
