"""TREAD token routing on the st355 path (reference: simpletuner/helpers/training/tread.py:58-159, wired into the transformers at
sd3/transformer.py:694-706, 796-803 and flux/transformer.py:1211-1241, 1394-1486).

A route {selection_ratio, start_layer_idx, end_layer_idx} shortens the IMAGE token sequence between two blocks: at the start block every sample keeps a
random subset of its tokens (a per-sample permutation; `force_keep` tokens always stay), the blocks in between run on the shorter sequence, and at the end
block the processed tokens go back to their slots while the skipped tokens re-enter with the values they had before the route — so the loss, and the
gradients, still cover every token.  On this path the two data movements are row gathers / scatters (st355_gather_rows / st355_scatter_rows), the router's
decision is B x S scalars of host-side plumbing (torch RNG on the device: distribution parity with the reference's CPU generator, not stream parity; tests
replay recorded permutations through `ReplayRouter`)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Optional

import torch


@dataclass
class MaskInfo:
    """tread.py:7-15"""
    mask: torch.Tensor            # [B, S] bool, True where the token was dropped
    ids_keep: torch.Tensor        # [B, K]
    ids_mask: torch.Tensor        # [B, S - K]
    ids_shuffle: torch.Tensor     # [B, S] permutation that packs the kept tokens first
    ids_restore: torch.Tensor     # inverse permutation

    def keep_i32(self) -> torch.Tensor:
        t = getattr(self, "_keep_i32", None)
        if t is None:
            t = self._keep_i32 = self.ids_keep.to(torch.int32).contiguous()
        return t


class TREADRouter:
    """same constructor and public API as the reference class (tread.py:18-159)"""

    def __init__(self, seed: int = 42, device: Any = None):
        self.generator = torch.Generator(device=device)
        self.generator.manual_seed(seed)

    @staticmethod
    def _importance(x: torch.Tensor) -> torch.Tensor:
        mags = x.float().abs().sum(-1)
        lo, hi = mags.min(dim=1, keepdim=True)[0], mags.max(dim=1, keepdim=True)[0]
        return (mags - lo) / (hi - lo + 1e-8)

    @torch.no_grad()
    def get_mask(self, x: torch.Tensor, mask_ratio: float = 0.0, l1_reg: float = 0.0, inverse: bool = False, force_keep: Optional[torch.Tensor] = None) -> MaskInfo:
        """tread.py:58-116.  x [B, S, D] (only its shape and device are read when l1_reg == 0, the reference's call sites' setting)"""
        B, S = x.shape[0], x.shape[1]
        dev = x.device
        if force_keep is None:
            force_keep = torch.zeros(B, S, dtype=torch.bool, device=dev)
        base_keep = S - int(round(S * float(mask_ratio)))
        K = max(base_keep, int(force_keep.sum(1).max()))
        noise = torch.rand(B, S, dtype=torch.float32, device=dev, generator=self.generator)
        mix = noise
        if l1_reg != 0.0:
            score = self._importance(x)
            if inverse:
                score = 1.0 - score
            mix = (1.0 - l1_reg) * noise + l1_reg * score
        mix = mix.masked_fill(force_keep, -1.0)
        ids_shuffle = torch.argsort(mix, dim=1)
        ids_keep, ids_mask = ids_shuffle[:, :K], ids_shuffle[:, K:]
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        mask = torch.ones(B, S, dtype=torch.bool, device=dev)
        mask.scatter_(1, ids_keep, False)
        return MaskInfo(mask, ids_keep, ids_mask, ids_shuffle, ids_restore)

    def start_route(self, x: torch.Tensor, info: MaskInfo) -> torch.Tensor:
        """tread.py:118-125: [B, S, D] -> [B, K, D], kept tokens in the permutation's order"""
        from .. import ops
        return ops.gather_rows(x, info.keep_i32())

    def end_route(self, routed_x: torch.Tensor, info: MaskInfo, original_x: Optional[torch.Tensor] = None, mask_token: float = 0.0) -> torch.Tensor:
        """tread.py:127-159: a full-length sequence again; skipped tokens carry `original_x` (or `mask_token`)"""
        from .. import ops
        full = original_x.clone() if original_x is not None else torch.full((routed_x.shape[0], info.mask.shape[1], routed_x.shape[2]), mask_token,
                                                                            dtype=routed_x.dtype, device=routed_x.device)
        return ops.scatter_rows(routed_x, info.keep_i32(), full)


class ReplayRouter(TREADRouter):
    """hands out recorded permutations (tests: the MaskInfo the REFERENCE router produced while its model files were executed)"""

    def __init__(self, infos: List[dict], device=None):
        self.infos, self.ptr, self.device = list(infos), 0, device

    def get_mask(self, x, mask_ratio: float = 0.0, l1_reg: float = 0.0, inverse: bool = False, force_keep=None) -> MaskInfo:
        d = self.infos[self.ptr % len(self.infos)]
        self.ptr += 1
        dev = x.device
        return MaskInfo(*(d[k].to(dev) for k in ("mask", "ids_keep", "ids_mask", "ids_shuffle", "ids_restore")))


def normalise_routes(routes, total_layers: int):
    """negative layer indices count from the end (sd3/transformer.py:708-721)"""
    out = []
    for r in routes or []:
        out.append(dict(r, start_layer_idx=r["start_layer_idx"] % total_layers if r["start_layer_idx"] < 0 else r["start_layer_idx"],
                        end_layer_idx=r["end_layer_idx"] % total_layers if r["end_layer_idx"] < 0 else r["end_layer_idx"]))
    return out
