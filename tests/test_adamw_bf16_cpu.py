"""AdamWBF16 (the examples' default optimizer) — the oracle restatement pinned BIT FOR BIT to the reference class executed in this
container (tools/gen_golden.py::gen_adamw_bf16 -> tests/golden/adamw_bf16_vectors.pt: 4 steps, two tensors, every stochastic-
rounding draw recorded)."""
from pathlib import Path

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
