"""Trainer.save_state / load_state: the file set and formats of the reference's checkpoints (accelerate.save_state names + the save hooks' files,
save_hooks.py:368-443, 850-1305; StateTracker.save_training_state keys, state_tracker.py:319-329).  Host-side round trip on the CPU (no optimizer
step is taken: stepping is a HIP launch — bit-exact optimizer / EMA resume on the MI355X is tests/test_optimizer_state_gpu.py)."""
import json
import pickle
from types import SimpleNamespace

import torch

from simpletuner_amd.foundation import ModelFoundation
from simpletuner_amd.training.trainer import Trainer, default_config


class _Comp(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.flat = (torch.randn(2 * 8 * 2, generator=g) * 0.1).to(torch.bfloat16)
        mk = lambda t: torch.nn.ParameterDict({"weight": torch.nn.Parameter(t)})
        self.blk = torch.nn.ModuleDict({"to_q": torch.nn.ModuleDict({"lora_A": torch.nn.ModuleDict({"default": mk(self.flat[:16].view(2, 8))}),
                                                                        "lora_B": torch.nn.ModuleDict({"default": mk(self.flat[16:].view(8, 2))})})})

    def trainable_parameters(self):
        return [self.blk["to_q"]["lora_A"]["default"]["weight"], self.blk["to_q"]["lora_B"]["default"]["weight"]]


class _Plug(ModelFoundation):
    MODEL_SUBFOLDER = "transformer"


def _trainer(seed, **cfg):
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    c = default_config(optimizer="adamw_bf16", use_ema=True, lr_scheduler="sine", lr_warmup_steps=5, lr_end=1e-6, lora_rank=2,
                       flow_custom_timesteps="100,200,300", flow_timesteps_mode="round-robin", **cfg)
    plug = _Plug(c, acc)
    plug.model = _Comp(seed)
    return Trainer(c, plug, acc), plug


def test_checkpoint_file_set_formats_and_roundtrip(tmp_path):
    tr, plug = _trainer(1)
    # put recognisable content into every piece of state without stepping
    st = tr.optimizer._init_group(0, tr.optimizer.param_groups[0])
    st["m"].copy_(torch.arange(32).to(torch.bfloat16) * 0.01); st["v"].fill_(0.5); st["shift"].fill_(-0.25); st["step"] = 6
    for i, p in enumerate(tr.params):
        tr.optimizer.state[p]["step"] = 6.0
        tr.optimizer.state[p]["accumulated_decay"] = 1e-3 * (i + 1)
    for _ in range(6):
        tr.lr_scheduler.step()
    tr.ema_model.shadow_flat.fill_(0.125); tr.ema_model.optimization_step = 6
    tr.state.update(global_step=6, micro_step=6, epoch=2, epoch_step=3)
    plug._noise_step = 6
    plug._noise_offset = 6 * 4099
    plug.sample_flow_sigmas({"latents": torch.zeros(2, 1, 2, 2)}, state={"global_step": 0})      # advances the round-robin cursor to 2
    torch.manual_seed(123)
    ck = tmp_path / "checkpoint-6"
    tr.save_state(str(ck))
    names = sorted(p.relative_to(ck).as_posix() for p in ck.rglob("*") if p.is_file())
    assert names == ["flow_custom_timestep_state.json", "optimizer.bin", "pytorch_lora_weights.safetensors", "random_states_0.pkl", "scheduler.bin",
                     "training_state.json", "transformer_ema/ema_model.pt"]
    ts = json.loads((ck / "training_state.json").read_text())
    want_ts = {"global_step": 6, "epoch_step": 3, "epoch": 2, "exhausted_backends": [], "repeats": {}}
    assert {k: ts[k] for k in want_ts} == want_ts
    opt_sd = torch.load(ck / "optimizer.bin", weights_only=True)            # no pickle execution needed for any file of the set
    assert set(opt_sd) == {"state", "param_groups"} and set(opt_sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq", "shift", "accumulated_decay"}
    from accelerate.utils.other import load as accelerate_load          # accelerate's own reader of its RNG file (weights_only + numpy allow-list)
    rng = accelerate_load(ck / "random_states_0.pkl")
    assert {"random_state", "numpy_random_seed", "torch_manual_seed", "torch_cuda_manual_seed"} <= set(rng) and rng["st355_noise_step"] == 6 and rng["st355_noise_offset"] == 6 * 4099
    want_draw = torch.rand(3)                                                   # what the run would have drawn next

    tr2, plug2 = _trainer(2)                                                    # a fresh process: different init everywhere
    torch.manual_seed(999)
    tr2.load_state(str(ck))
    assert torch.equal(plug2.model.flat, plug.model.flat)                       # adapter weights
    f1, f2 = tr.optimizer._flat[0], tr2.optimizer._flat[0]
    assert f2["step"] == 6 and all(torch.equal(f1[k], f2[k]) for k in ("m", "v", "shift"))
    assert [tr2.optimizer.state[p]["accumulated_decay"] for p in tr2.params] == [1e-3, 2e-3]
    assert tr2.lr_scheduler.last_epoch == tr.lr_scheduler.last_epoch == 6 and tr2.optimizer.param_groups[0]["lr"] == tr.optimizer.param_groups[0]["lr"]
    assert torch.equal(tr2.ema_model.shadow_flat, tr.ema_model.shadow_flat) and tr2.ema_model.optimization_step == 6
    assert tr2.state["global_step"] == 6 and tr2.state["micro_step"] == 6 and tr2.state["epoch"] == 2 and plug2._noise_step == 6 and plug2._noise_offset == 6 * 4099
    assert torch.equal(torch.rand(3), want_draw)                                # the torch RNG stream continues
    _, t = plug2.sample_flow_sigmas({"latents": torch.zeros(2, 1, 2, 2)}, state={"global_step": 6})
    assert torch.equal(t, torch.tensor([300.0, 100.0]))                         # the cursor continues at 2


def test_deferred_non_finite_loss_check_raises_the_reference_error():
    """trainer.py:7102-7110 semantics, deferred by one step (Trainer.check_pending_loss): finite -> silent and cleared, NaN / inf -> RuntimeError
    naming the loss, the data backend and the file paths"""
    import pytest
    tr = Trainer.__new__(Trainer)
    tr._pending_loss = (torch.tensor(0.25), 3, ["a.png"], "ds1")
    tr.check_pending_loss()
    assert tr._pending_loss is None
    tr.check_pending_loss()                                                     # nothing pending: no-op
    tr._pending_loss = (torch.tensor(float("nan")), 4, ["b.png", "c.png"], "ds1")
    with pytest.raises(RuntimeError, match=r"Non-finite training loss detected \(loss=nan, data_backend_id=ds1, filepaths=\['b.png', 'c.png'\]"):
        tr.check_pending_loss()
    assert tr._pending_loss is None


def test_unknown_optimizer_name_is_refused():
    import pytest
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    c = default_config(optimizer="lion")
    plug = _Plug(c, acc)
    plug.model = _Comp(1)
    with pytest.raises(NotImplementedError, match="optimizer 'lion'"):
        Trainer(c, plug, acc)
