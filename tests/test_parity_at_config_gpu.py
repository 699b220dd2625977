"""Parity at the BASELINE.json configurations other than Flux: see tests/parity_at_config.py — the same functions bench.py reports as
`parity_at_config` on `--model sd15 / sdxl / sd3 --full / pixart`.  (Flux: tests/test_baseline_shapes_gpu.py, full width AND full depth.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import parity_at_config as PC  # noqa: E402

DEV = torch.device("cuda:0")


def _check(rep):
    print("[parity@config]", {k: v for k, v in rep.items() if k != "tolerance"})
    assert rep["pred_rel_l2"] < 2e-2 and rep["pred_cos"] > 0.9995, rep
    assert rep["grad_worst_vs_its_tolerance"] < 1.0 and rep["grad_worst_cos"] > 0.995, rep
    assert rep["grads_compared"] > 20, rep


def test_sd15_lora_r16_512_true_architecture():
    """BASELINE.json configs[0]: SD 1.5 UNet LoRA rank 16, 512^2, batch 1"""
    _check(PC.unet("sd15", 512, DEV, lora=True, rank=16))


@pytest.mark.parametrize("lora", [False, True], ids=["full_finetune", "lora_r16"])
def test_sdxl_1024_true_architecture(lora):
    """BASELINE.json configs[1] (SDXL UNet full fine-tune bf16, 1024^2) and the metric's SDXL-LoRA"""
    _check(PC.unet("sdxl", 1024, DEV, lora=lora, rank=16))


def test_sd3_medium_full_finetune_1024_true_width():
    """BASELINE.json configs[3]: SD3-Medium MMDiT full fine-tune, 1024^2 (2 of the 24 joint blocks: the quick form)"""
    rep = PC.sd3_full(1024, DEV)
    _check(rep)
    assert abs(rep["loss_hip"] - rep["loss_oracle"]) < 1e-3 * max(1.0, abs(rep["loss_oracle"]))


def test_sd3_medium_full_finetune_full_depth_on_a_mixed_aspect_bucket():
    """BASELINE.json configs[3] at its real depth: all 24 joint blocks, full fine-tune, the 1216 x 832 bucket (S = 3952 + 231: a ragged last key tile)"""
    rep = PC.sd3_full(1024, DEV, layers=24, hw=(1216, 832))
    _check(rep)
    assert abs(rep["loss_hip"] - rep["loss_oracle"]) < 1e-3 * max(1.0, abs(rep["loss_oracle"]))


def test_pixart_sigma_controlnet_2k_true_width():
    """BASELINE.json configs[4]: PixArt-Sigma ControlNet branch, 2K latents (S=16384), T5 context 300 with mask (3 trunk + 2 adapter blocks: the quick form)"""
    _check(PC.pixart_controlnet(2048, DEV))


def test_pixart_sigma_controlnet_2k_full_depth():
    """BASELINE.json configs[4] at its real depth: 28 trunk blocks + 13 ControlNet blocks at 2K latents"""
    _check(PC.pixart_controlnet(2048, DEV, trunk_layers=28, ctrl_layers=13))
