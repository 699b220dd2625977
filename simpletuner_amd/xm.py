"""Explorative modelling with noise candidates ("XM", off by default): the plugin-side mirror of the reference's
simpletuner/helpers/models/xm_mixin.py:16-485 and simpletuner/helpers/training/explorative_modeling.py:13-91, for the noise-candidate mode the
in-scope families support (`xm_training_target='noise'`, `xm_selection_scope='sample'`).

Per step: every sample of the prepared batch is replicated K times CANDIDATE-MAJOR (row k*B + b = candidate k of sample b), each replica gets
its own noise draw at the SAME sigma / timestep, the network runs once on the K*B batch, and only the candidate with the smallest per-sample
loss trains:  loss = mean_b min_k L[k, b].

MI355X form: the K*B forward is one larger batch through the same kernels.  The per-sample losses come from the fused loss kernel run once
WITHOUT its gradient output (4 B/element); the winners are picked on the device (argmin over a [K, B] view — no host sync); the training loss
and d(loss)/d(pred) then come from the same kernel run with per-sample weights  K * [k == winner_b]  (times the min-SNR weight when one
applies), so mean over K*B rows equals the mean over the B winners and losing candidates receive an exactly-zero gradient.  Batch / output
dictionaries are then cut back to the B winning rows, as the reference does, so auxiliary losses and logging see the original batch size.
"""
from __future__ import annotations

from dataclasses import dataclass, fields, is_dataclass, replace
from typing import Optional, Tuple

import torch


@dataclass
class ExplorativeModelingConfig:
    """explorative_modeling.py:13-56"""
    enabled: bool
    candidate_count: int
    training_target: str
    selection_scope: str
    block_size: int

    @classmethod
    def from_config(cls, config) -> "ExplorativeModelingConfig":
        def opt(name, default):
            if isinstance(config, dict):
                return config.get(name, default)
            try:
                return vars(config).get(name, default)
            except TypeError:
                return getattr(config, name, default)

        out = cls(enabled=bool(opt("xm_enabled", False)), candidate_count=int(opt("xm_candidate_count", 1) or 1),
                  training_target=str(opt("xm_training_target", "noise") or "noise"), selection_scope=str(opt("xm_selection_scope", "sample") or "sample"),
                  block_size=int(opt("xm_block_size", 0) or 0))
        if out.training_target not in ("noise", "route"):
            raise ValueError("xm_training_target must be 'noise' or 'route'.")
        if out.selection_scope not in ("sample", "block"):
            raise ValueError("xm_selection_scope must be 'sample' or 'block'.")
        if out.enabled and out.candidate_count < 2:
            raise ValueError("xm_candidate_count must be at least 2 when XM is enabled.")
        if out.block_size < 0:
            raise ValueError("xm_block_size must be non-negative.")
        if out.selection_scope == "block" and out.block_size == 1:
            raise ValueError("xm_block_size=1 would select winners per token; use sample scope or a larger block.")
        return out


# ---- candidate-major batch algebra (explorative_modeling.py:59-107, 156-159) ----
def reduce_loss_to_samples(loss: torch.Tensor) -> torch.Tensor:
    return loss.reshape(1) if loss.ndim == 0 else loss.float().mean(dim=tuple(range(1, loss.ndim)))


def reshape_candidate_batch(value: torch.Tensor, candidate_count: int) -> torch.Tensor:
    if candidate_count < 1:
        raise ValueError("candidate_count must be positive.")
    if value.shape[0] % candidate_count != 0:
        raise ValueError(f"Tensor batch dimension {value.shape[0]} is not divisible by candidate_count={candidate_count}.")
    return value.reshape(candidate_count, value.shape[0] // candidate_count, *value.shape[1:])


def select_min_candidate_loss(candidate_losses: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    if candidate_losses.ndim != 2:
        raise ValueError(f"XM candidate losses must have shape [candidates, batch], got {tuple(candidate_losses.shape)}.")
    best, who = candidate_losses.min(dim=0)
    return best.mean(), who


def select_winning_candidates(value: torch.Tensor, winner_indices: torch.Tensor, candidate_count: int) -> torch.Tensor:
    view = reshape_candidate_batch(value, candidate_count)
    if winner_indices.ndim != 1 or winner_indices.shape[0] != view.shape[1]:
        raise ValueError("winner_indices must have shape [batch] matching the candidate-expanded tensor's original batch size.")
    return view[winner_indices.to(device=value.device, dtype=torch.long), torch.arange(view.shape[1], device=value.device)]


def route_usage_histogram(winner_indices: torch.Tensor, candidate_count: int) -> Optional[torch.Tensor]:
    if winner_indices.numel() == 0:
        return None
    return torch.bincount(winner_indices.to(dtype=torch.long), minlength=candidate_count).float()


def winner_weights(winner_indices: torch.Tensor, candidate_count: int, base: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[K*B] fp32 row weights: K on each sample's winning candidate row, 0 elsewhere (times `base`): the mean over K*B weighted rows is the
    mean over the B winners"""
    B = winner_indices.shape[0]
    w = torch.zeros(candidate_count, B, dtype=torch.float32, device=winner_indices.device)
    w.scatter_(0, winner_indices.reshape(1, B).long(), float(candidate_count))
    w = w.reshape(-1)
    return w if base is None else w * base.to(w)


def _map_batch_rows(value, fn_tensor, fn_seq, rows: int, key=None, keep_lists=frozenset()):
    """walk tensors / lists / tuples / dicts / dataclasses; leaves whose leading dimension is `rows` go through fn_tensor (fn_seq for python
    sequences of that length), everything else is returned as is (xm_mixin.py:75-92, 327-388)"""
    if torch.is_tensor(value):
        return fn_tensor(value) if value.ndim > 0 and value.shape[0] == rows else value
    if isinstance(value, (list, tuple)):
        if fn_seq is not None and key not in keep_lists and len(value) == rows:
            out = fn_seq(list(value))
        else:
            out = [_map_batch_rows(v, fn_tensor, fn_seq, rows, key, keep_lists) for v in value]
        return tuple(out) if isinstance(value, tuple) else out
    if isinstance(value, dict):
        return {k: _map_batch_rows(v, fn_tensor, fn_seq, rows, k, keep_lists) for k, v in value.items()}
    if is_dataclass(value) and not isinstance(value, type):
        return replace(value, **{f.name: _map_batch_rows(getattr(value, f.name), fn_tensor, fn_seq, rows, f.name, keep_lists) for f in fields(value)})
    return value


class ExplorativeModelingMixin:
    XM_SEQUENCE_LIST_KEYS = frozenset()

    def _prediction_type_value(self) -> str:
        return str(getattr(self.PREDICTION_TYPE, "value", self.PREDICTION_TYPE))

    def _validate_xm_support(self) -> None:
        """xm_mixin.py:21-57: the noise-candidate mode excludes every option that re-draws or re-times the noise behind its back"""
        xm = getattr(self, "xm_config", None)
        if xm is None or not xm.enabled:
            return
        name, cfg = self.NAME, self.config
        if xm.training_target != "noise":
            raise ValueError(f"{name} XM currently supports only xm_training_target='noise'.")
        if xm.selection_scope != "sample":
            raise ValueError(f"{name} XM noise-candidate training requires xm_selection_scope='sample'.")
        if int(getattr(xm, "block_size", 0) or 0) != 0:
            raise ValueError(f"{name} XM noise-candidate training requires xm_block_size=0.")
        if getattr(cfg, "twinflow_enabled", False):
            raise ValueError(f"{name} XM noise-candidate training is not compatible with TwinFlow.")
        if getattr(cfg, "scheduled_sampling_reflexflow", False) or int(getattr(cfg, "scheduled_sampling_max_step_offset", 0) or 0) > 0:
            raise ValueError(f"{name} XM noise-candidate training is not compatible with scheduled sampling.")
        if float(getattr(cfg, "input_perturbation", 0.0) or 0.0) != 0.0:
            raise ValueError(f"{name} XM noise-candidate training is not compatible with input_perturbation.")
        if getattr(cfg, "crepa_self_flow", False) or getattr(cfg, "crepa_feature_source", None) == "self_flow":
            raise ValueError(f"{name} XM noise-candidate training is not compatible with CREPA self-flow.")

    def _xm_noise_candidates_enabled(self, prepared_batch: Optional[dict] = None) -> bool:
        xm = getattr(self, "xm_config", None)
        if not xm or not xm.enabled:
            return False
        self._validate_xm_support()
        if prepared_batch is not None and (prepared_batch.get("xm_candidate_count") or prepared_batch.get("xm_winner_indices") is not None):
            return False                                                    # already expanded / already cut back this step
        return xm.training_target == "noise"

    def _repeat_xm_candidate_value(self, value, candidate_count: int, batch_size: int):
        """tensors AND python sequences with one entry per sample are repeated candidate-major (sd3/model.py:467-488; the generic mixin leaves
        python lists alone, which is equivalent as long as nobody indexes them with an expanded row number — repeating is the safe reading)"""
        return _map_batch_rows(value, lambda t: t.repeat((candidate_count,) + (1,) * (t.ndim - 1)), lambda seq: seq * candidate_count, batch_size)

    def _xm_noise_mix(self, latents, noise, batch):
        """(noisy_latents, extra keys) for the expanded batch — on the HIP noising kernels, with the candidates' noise handed in"""
        from . import ops
        kind = self._prediction_type_value()
        if kind == "flow_matching":
            sig = batch.get("mixflow_interpolation_sigmas")
            if sig is None:
                sig = batch.get("sigmas")
            if not torch.is_tensor(sig):
                raise ValueError(f"{self.NAME} XM noise-candidate training requires tensor sigmas for flow interpolation.")
            flat = sig.reshape(sig.shape[0], -1)
            if flat.shape[1] > 1 and not torch.allclose(flat, flat[:, :1].expand_as(flat)):          # sd3/model.py:512-514
                raise ValueError(f"{self.NAME} XM noise-candidate training requires per-sample scalar sigmas.")
            sig = flat[:, 0].to(device=latents.device, dtype=torch.float32).contiguous()
            noisy, target, _ = ops.flow_noise_mix(latents, sig, noise=noise)
            return noisy, {"flow_target": target}
        if kind in ("epsilon", "v_prediction"):
            a, b = self.noise_schedule.mix_coefficients(batch["timesteps"])
            noisy, vel = ops.ddpm_noise_mix(latents, noise, a, b, want_v=(kind == "v_prediction"))
            return noisy, ({"velocity_target": vel} if kind == "v_prediction" else {})
        raise ValueError(f"{self.NAME} XM noise-candidate training does not support {kind}.")

    def _prepare_xm_noise_candidates(self, prepared_batch: dict) -> dict:
        """xm_mixin.py:94-158 (flux/model.py:638-680 is the same algebra): expand candidate-major, draw K*B noises, re-noise, re-target"""
        K = self.xm_config.candidate_count
        lat, ts = prepared_batch.get("latents"), prepared_batch.get("timesteps")
        if not torch.is_tensor(lat) or not torch.is_tensor(ts):
            raise ValueError(f"{self.NAME} XM noise-candidate training requires latents and timesteps tensors.")
        if "noisy_latents" not in prepared_batch:
            raise ValueError(f"{self.NAME} XM noise-candidate training requires prepared noisy_latents.")
        if prepared_batch.get("target") is not None:
            raise ValueError(f"{self.NAME} XM noise-candidate training cannot be used with an explicit prepared target.")
        B = lat.shape[0]
        big = {k: self._repeat_xm_candidate_value(v, K, B) for k, v in prepared_batch.items()}
        noise = torch.randn_like(big["latents"])
        big["noise"] = big["input_noise"] = noise
        big["noisy_latents"], extra = self._xm_noise_mix(big["latents"].contiguous(), noise, big)
        big.update(extra)
        big["xm_candidate_count"], big["xm_original_batch_size"] = K, B
        prepared_batch.clear()
        prepared_batch.update(big)
        return prepared_batch

    def _select_xm_winners_in_place(self, prepared_batch: dict, model_output: dict, winner_indices: torch.Tensor, candidate_count: int) -> None:
        """xm_mixin.py:293-325: both dictionaries shrink to the B winning rows"""
        B = int(winner_indices.shape[0])
        rows = candidate_count * B

        def pick_rows(t):
            return select_winning_candidates(t, winner_indices, candidate_count)

        def pick_items(seq):
            who = winner_indices.detach().to(device="cpu", dtype=torch.long).tolist()
            return [seq[int(k) * B + b] for b, k in enumerate(who)]

        keep = getattr(self, "XM_SEQUENCE_LIST_KEYS", frozenset())
        for k in list(prepared_batch):
            prepared_batch[k] = _map_batch_rows(prepared_batch[k], pick_rows, pick_items, rows, k, keep)
        prepared_batch.pop("xm_candidate_count", None)
        prepared_batch.pop("xm_original_batch_size", None)
        prepared_batch["xm_winner_indices"] = winner_indices.detach()
        for k in list(model_output):
            if k not in ("xm_candidate_count", "xm_winner_indices"):
                model_output[k] = _map_batch_rows(model_output[k], pick_rows, pick_items, rows, k, keep)
        model_output["xm_winner_indices"] = winner_indices.detach()
        model_output.pop("xm_candidate_count", None)

    @staticmethod
    def _xm_candidate_logs(selected_loss, candidate_losses, winner_indices, candidate_count: int) -> dict:
        logs = {"xm_loss": selected_loss.detach().item(), "xm_candidate_loss_mean": candidate_losses.detach().float().mean().item()}
        usage = route_usage_histogram(winner_indices, candidate_count)
        if usage is not None:
            for i, n in enumerate(usage.to("cpu").tolist()):
                logs[f"xm_candidate_{i}_wins"] = n
        return logs

    def _xm_noise_loss_with_logs(self, prepared_batch: dict, model_output: dict, *, candidate_count: int, apply_conditioning_mask: bool = True):
        """xm_mixin.py:448-474.  Two runs of the fused loss kernel: (1) per-row losses, no gradient; (2) loss + gradient with winner weights."""
        if candidate_count < 2:
            raise ValueError(f"{self.NAME} XM candidate_count must be at least 2.")
        per_row, base_w = self.loss(prepared_batch, model_output, apply_conditioning_mask=apply_conditioning_mask, _per_sample_only=True)
        cand = reshape_candidate_batch(per_row, candidate_count)
        _, who = select_min_candidate_loss(cand)
        # the weighted per-row losses already carry base_w; the second pass needs base_w * K * [winner]
        loss = self.loss(prepared_batch, model_output, apply_conditioning_mask=apply_conditioning_mask,
                         _row_weight=winner_weights(who, candidate_count, base_w))
        self._select_xm_winners_in_place(prepared_batch, model_output, who, candidate_count)
        return loss, self._xm_candidate_logs(loss, cand, who, candidate_count)
