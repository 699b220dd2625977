"""XM noise candidates' HOST logic on the CPU (reference: xm_mixin.py:448-485; tests/test_flux_model.py:132-165): the per-row losses come from the fused loss contract
(tests/ops_emulator.py), the candidate selection, batch slimming, logs and the zeroed loser gradients are the plugin's own code (simpletuner_amd/xm.py, foundation.py)."""
from types import SimpleNamespace

import pytest
import torch

from tests import ops_emulator as EMU

BF16 = torch.bfloat16


def test_xm_loss_selects_winners_and_zeroes_loser_gradients(monkeypatch):
    EMU.install(monkeypatch)
    from simpletuner_amd.sdxl.model import SDXL
    from simpletuner_amd.training.trainer import default_config
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    pl = SDXL(default_config(model_family="sdxl", model_type="lora", xm_enabled=True, xm_candidate_count=2), acc)
    pl.setup_training_noise_schedule()
    noise = torch.tensor([0.0, 1.0, 2.0, 3.0]).view(4, 1, 1, 1).expand(4, 1, 2, 8).contiguous().to(BF16)
    pred0 = torch.tensor([5.0, 1.5, 2.25, -4.0]).view(4, 1, 1, 1).expand(4, 1, 2, 8).contiguous().to(BF16)
    pred = pred0.clone().requires_grad_(True)
    hidden = torch.arange(4 * 3 * 2, dtype=torch.float32).reshape(4, 3, 2)
    pb = {"latents": torch.zeros(4, 1, 2, 8, dtype=BF16), "noise": noise, "timesteps": torch.tensor([100, 200, 100, 200]),
          "metadata": [{"id": 0}, {"id": 1}, {"id": 0}, {"id": 1}], "xm_candidate_count": 2, "xm_original_batch_size": 2}
    out = {"model_prediction": pred, "hidden_states_buffer": {"layer_2": hidden.clone()}, "xm_candidate_count": 2}
    loss, logs = pl.loss_with_logs(pb, out)
    loss.backward()
    # per-row losses [25, .25, .0625, 49] -> candidates [[25, .25], [.0625, 49]] -> winners [1, 0]
    assert out["xm_winner_indices"].tolist() == [1, 0]
    assert loss.item() == pytest.approx((0.0625 + 0.25) / 2, rel=1e-5)
    assert logs["xm_loss"] == pytest.approx(loss.item()) and logs["xm_candidate_loss_mean"] == pytest.approx((25 + 0.25 + 0.0625 + 49) / 4, rel=1e-5)
    assert logs["xm_candidate_0_wins"] == 1.0 and logs["xm_candidate_1_wins"] == 1.0
    assert pb["latents"].shape[0] == 2 and pb["metadata"] == [{"id": 0}, {"id": 1}] and "xm_candidate_count" not in pb and "xm_candidate_count" not in out
    assert out["model_prediction"].shape[0] == 2 and torch.equal(out["hidden_states_buffer"]["layer_2"], hidden[[2, 1]])
    g = pred.grad.float()
    assert torch.count_nonzero(g[0]) == 0 and torch.count_nonzero(g[3]) == 0
    want = 2.0 * (pred0.float() - noise.float()) / (16 * 2)               # d/dpred of the mean over the 2 winners of their 16-element means
    assert torch.allclose(g[1], want[1], rtol=1e-2) and torch.allclose(g[2], want[2], rtol=1e-2)
    # the reference's own case (tests/test_flux_model.py:132-160): the winners predict their targets exactly -> loss 0, winners [1, 0]
    pb2 = {"latents": torch.zeros(4, 1, 2, 8, dtype=BF16), "noise": noise, "timesteps": torch.tensor([100, 200, 100, 200])}
    out2 = {"model_prediction": torch.tensor([5.0, 1.0, 2.0, -4.0]).view(4, 1, 1, 1).expand(4, 1, 2, 8).contiguous().to(BF16), "xm_candidate_count": 2}
    l2, _ = pl.loss_with_logs(pb2, out2)
    assert l2.item() == 0.0 and out2["xm_winner_indices"].tolist() == [1, 0]
