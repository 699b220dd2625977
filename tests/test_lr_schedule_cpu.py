"""Learning-rate schedules (simpletuner_amd/training/lr_schedule.py) against the reference's scheduler classes stepped on a torch optimizer
(tools/gen_golden.py::gen_lr_schedules -> tests/golden/lr_schedule_vectors.pt): every rate of 45 steps, exactly."""
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch

from simpletuner_amd.training.lr_schedule import get_lr_scheduler

G = torch.load(Path(__file__).parent / "golden" / "lr_schedule_vectors.pt", weights_only=False)


def _run(args, n, world=1, global_step=0):
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch = get_lr_scheduler(SimpleNamespace(**args), opt, SimpleNamespace(num_processes=world), None, global_step)
    out = [opt.param_groups[0]["lr"]]
    for _ in range(n):
        sch.step()
        out.append(opt.param_groups[0]["lr"])
    return out, sch, opt


def test_periodic_schedules_match_reference_classes_exactly():
    seen = 0
    for key, want in G.items():
        if key[0] == "polynomial":
            continue
        name, T0, eta = key
        got, _, _ = _run(dict(lr_scheduler=name, lr_warmup_steps=T0, lr_end=eta), len(want) - 1)
        assert got == want, (key, [i for i, (a, b) in enumerate(zip(got, want)) if a != b][:5])
        seen += 1
    assert seen == 9
    # the documented shapes: sine starts at the midpoint and peaks at T_0/2; the reference's "cosine_with_restarts" never moves
    sine = G[("sine", 10, 0.0)]
    assert sine[0] == pytest.approx(5e-5) and max(sine) == sine[5] == pytest.approx(1e-4) and min(sine) == sine[15] == 0.0
    assert set(G[("cosine_with_restarts", 10, 0.0)]) == {1e-4}
    cos = G[("cosine", 10, 0.0)]
    assert cos[0] == pytest.approx(1e-4, rel=1e-6) and cos[10] == pytest.approx(0.0, abs=1e-12) and cos[20] == pytest.approx(1e-4, rel=1e-6)


def test_polynomial_matches_reference_function_and_scales_with_world():
    for key, want in G.items():
        if key[0] != "polynomial":
            continue
        _, warm, total, end, power = key
        got, _, _ = _run(dict(lr_scheduler="polynomial", lr_warmup_steps=warm, max_train_steps=total, lr_end=end, lr_power=power), len(want) - 1)
        assert got == pytest.approx(want, rel=1e-12, abs=0), key
    # warm-up and horizon are multiplied by the process count (custom_schedule.py:537-544)
    two, _, _ = _run(dict(lr_scheduler="polynomial", lr_warmup_steps=5, max_train_steps=15, lr_end=1e-7, lr_power=1.0), 12, world=2)
    assert two[10] == pytest.approx(1e-4) and two[5] == pytest.approx(5e-5)
    with pytest.raises(ValueError, match="must be be smaller than initial lr"):
        _run(dict(lr_scheduler="polynomial", lr_warmup_steps=1, max_train_steps=5, lr_end=1.0), 1)


def test_resume_and_generic_names():
    full, _, _ = _run(dict(lr_scheduler="sine", lr_warmup_steps=7, lr_end=1e-6), 20)
    part, sch, opt = _run(dict(lr_scheduler="sine", lr_warmup_steps=7, lr_end=1e-6), 8)
    sd = sch.state_dict()
    opt2 = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch2 = get_lr_scheduler(SimpleNamespace(lr_scheduler="sine", lr_warmup_steps=7, lr_end=1e-6), opt2, SimpleNamespace(num_processes=1))
    sch2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == part[-1] == full[8]
    rest = []
    for _ in range(12):
        sch2.step()
        rest.append(opt2.param_groups[0]["lr"])
    assert rest == full[9:] and sch2.get_last_lr() == [full[-1]]
    # polynomial resumed by construction at global_step (the reference passes last_epoch = global_step - 1)
    ref, _, _ = _run(dict(lr_scheduler="polynomial", lr_warmup_steps=5, max_train_steps=30, lr_end=1e-7, lr_power=1.0), 20)
    res, _, _ = _run(dict(lr_scheduler="polynomial", lr_warmup_steps=5, max_train_steps=30, lr_end=1e-7, lr_power=1.0), 8, global_step=12)
    assert res == pytest.approx(ref[12:21], rel=1e-12)
    const, _, _ = _run(dict(lr_scheduler="constant"), 3)
    warm, _, _ = _run(dict(lr_scheduler="constant_with_warmup", lr_warmup_steps=4), 6)
    assert const == [1e-4] * 4 and warm == pytest.approx([0.0, 2.5e-5, 5e-5, 7.5e-5, 1e-4, 1e-4, 1e-4])
    with pytest.raises(NotImplementedError):
        _run(dict(lr_scheduler="piecewise_constant"), 1)
    with pytest.raises(ValueError, match="positive integer T_0"):
        _run(dict(lr_scheduler="sine", lr_warmup_steps=0), 1)
