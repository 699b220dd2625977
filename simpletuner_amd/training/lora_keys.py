"""LoRA checkpoint key dialects — SURVEY.md §8(f)2: "write/read exactly the reference's LoRA safetensors keys (diffusers/ComfyUI formats)".

The reference converts between three spellings of the same two matrices per wrapped Linear (simpletuner/helpers/training/lora_format.py:1-375;
applied on save / load by common.py:1972, 2050-2120):

    dialect                     down (A, [r, in])              up (B, [out, r])             scale
    peft / diffusers (new)      <comp>.<mod>.lora_A.weight     <comp>.<mod>.lora_B.weight   adapter metadata (lora_alpha, alpha_pattern)
    diffusers (old)             <comp>.<mod>.lora.down.weight  <comp>.<mod>.lora.up.weight  network_alphas
    ComfyUI                     diffusion_model.<mod>.lora_A.weight / .lora_B.weight        <...>.alpha tensor per module
    ComfyUI, SD / SDXL (kohya)  lora_unet_<mod with _>.lora_down.weight / .lora_up.weight   <...>.alpha tensor per module

Pure dictionary work on the host (a LoRA file is a few hundred small tensors); pinned to the reference module executed as it lies
(tools/gen_golden.py::gen_lora_keys -> tests/golden/lora_keys_vectors.pt).
"""
from __future__ import annotations

from collections import Counter
from enum import Enum
from typing import Any, Dict, Iterable, Optional, Set, Tuple

import torch

DOWN_SUFFIXES = (".lora.down.weight", ".lora_A.weight", ".lora_down.weight")
UP_SUFFIXES = (".lora.up.weight", ".lora_B.weight", ".lora_up.weight")
ALPHA_SUFFIXES = (".alpha", ".lora_alpha")
COMPONENTS = ("unet.", "transformer.", "controlnet.")
KNOWN_PREFIXES = ("text_encoder.", "text_encoder_2.", "controlnet.", "unet.", "transformer.")
KOHYA_COMPONENT = {"unet": "lora_unet", "text_encoder_2": "lora_te2"}      # text_encoder: lora_te1 (SDXL) / lora_te (SD1.x, SD2.x)


class PEFTLoRAFormat(str, Enum):
    DIFFUSERS = "diffusers"
    COMFYUI = "comfyui"


def normalize_lora_format(value) -> PEFTLoRAFormat:
    """anything that is not (a spelling of) "comfyui" means diffusers (lora_format.py:17-28)"""
    if isinstance(value, PEFTLoRAFormat):
        return value
    if isinstance(value, str) and value.strip().lower() == PEFTLoRAFormat.COMFYUI.value:
        return PEFTLoRAFormat.COMFYUI
    return PEFTLoRAFormat.DIFFUSERS


def detect_state_dict_format(state_dict: Dict[str, Any]) -> Optional[PEFTLoRAFormat]:
    """lora_format.py:31-46: ComfyUI when keys carry its `diffusion_model.` prefix, or `.alpha` tensors appear without any old-diffusers
    `.lora.down/.lora.up` key"""
    if not state_dict:
        return None
    comfy_prefix = any(k.startswith(("diffusion_model.", "model.diffusion_model.")) for k in state_dict)
    has_alpha = any(k.endswith(".alpha") for k in state_dict)
    old_diffusers = any(".lora.down" in k or ".lora.up" in k for k in state_dict)
    return PEFTLoRAFormat.COMFYUI if comfy_prefix or (has_alpha and not old_diffusers) else PEFTLoRAFormat.DIFFUSERS


def _number(v) -> float:
    return float(v.detach().float().cpu().item()) if torch.is_tensor(v) else float(v)


def _without(key: str, prefix: Optional[str]) -> str:
    return key[len(prefix):] if prefix and key.startswith(prefix) else key


def _split_suffix(key: str, suffixes: Iterable[str]) -> Optional[Tuple[str, str]]:
    for suf in suffixes:
        if key.endswith(suf):
            return key[:-len(suf)], suf
    return None


def collect_lora_ranks(state_dict: Dict[str, Any], *, prefix_to_strip: Optional[str] = None) -> Dict[str, int]:
    """module -> rank, read off the down matrix's rows or the up matrix's columns; disagreeing halves are an error (lora_format.py:65-90)"""
    ranks: Dict[str, int] = {}
    for key, val in state_dict.items():
        if not hasattr(val, "shape"):
            continue
        k = _without(key, prefix_to_strip)
        hit, axis = _split_suffix(k, DOWN_SUFFIXES), 0
        if hit is None:
            hit, axis = _split_suffix(k, UP_SUFFIXES), 1
        if hit is None:
            continue
        module, r = hit[0], int(val.shape[axis])
        if ranks.setdefault(module, r) != r:
            raise ValueError(f"LoRA checkpoint has conflicting ranks for `{module}`: {ranks[module]} and {r}.")
    return ranks


def collect_lora_alphas(state_dict: Dict[str, Any], *, prefix_to_strip: Optional[str] = None) -> Dict[str, float]:
    out: Dict[str, float] = {}
    for key, val in state_dict.items():
        hit = _split_suffix(_without(key, prefix_to_strip), ALPHA_SUFFIXES)
        if hit is not None:
            out[hit[0]] = _number(val)
    return out


def synthesize_missing_lora_alphas_from_ranks(state_dict: Dict[str, Any], *, existing_alphas: Optional[Dict[str, Any]] = None,
                                              prefix_to_strip: Optional[str] = None) -> Dict[str, float]:
    """a mixed-rank file WITHOUT alpha data gets alpha = rank per module (scale 1); uniform-rank files keep the configured alpha (:104-122)"""
    if existing_alphas or collect_lora_alphas(state_dict, prefix_to_strip=prefix_to_strip):
        return {}
    ranks = collect_lora_ranks(state_dict, prefix_to_strip=prefix_to_strip)
    if len(set(ranks.values())) <= 1:
        return {}
    return {f"{m}.alpha": float(r) for m, r in ranks.items()}


def _mode(values):
    return max(Counter(values).items(), key=lambda kv: (kv[1], kv[0]))[0]      # most frequent; ties go to the larger value


def peft_lora_config_kwargs_from_state_dict(state_dict: Dict[str, Any], *, prefix_to_strip: Optional[str] = None) -> Dict[str, Any]:
    """LoraConfig(r=, lora_alpha=, rank_pattern=, alpha_pattern=) that reproduces the file's per-module ranks / alphas (:125-158)"""
    ranks = collect_lora_ranks(state_dict, prefix_to_strip=prefix_to_strip)
    if not ranks:
        return {}
    r0 = _mode(list(ranks.values()))
    alphas = collect_lora_alphas(state_dict, prefix_to_strip=prefix_to_strip)
    if alphas:
        a0 = _mode(list(alphas.values()))
        alpha_pattern = {m: a for m, a in alphas.items() if m in ranks and a != a0}
    else:
        a0 = float(r0)
        alpha_pattern = {m: float(r) for m, r in ranks.items() if r != r0}
    out: Dict[str, Any] = {"r": r0, "lora_alpha": a0}
    rank_pattern = {m: r for m, r in ranks.items() if r != r0}
    if rank_pattern:
        out["rank_pattern"] = rank_pattern
    if alpha_pattern:
        out["alpha_pattern"] = alpha_pattern
    return out


def convert_comfyui_to_diffusers(state_dict: Dict[str, Any], target_prefix: Optional[str] = None) -> Tuple[Dict[str, Any], Dict[str, float]]:
    """ComfyUI -> old-diffusers keys under `target_prefix` + {"<module>.alpha": value} (:175-215)"""
    pre = f"{target_prefix}." if target_prefix else ""
    out, alphas = {}, {}
    for key, val in state_dict.items():
        if key.startswith("diffusion_model."):
            key = pre + key[len("diffusion_model."):]
        elif pre and key.startswith(pre):
            pass
        elif target_prefix and not key.startswith(KNOWN_PREFIXES):
            key = pre + key
        if key.endswith(".alpha"):
            try:
                alphas[key] = _number(val)
            except Exception:
                pass
            continue
        if key.endswith(".lora_A.weight"):
            key = key[:-len(".lora_A.weight")] + ".lora.down.weight"
        elif key.endswith(".lora_B.weight"):
            key = key[:-len(".lora_B.weight")] + ".lora.up.weight"
        out[key] = val
    return out, alphas


def _alpha_for(module: str, down_weight, meta: Optional[dict]) -> Optional[float]:
    """alpha_pattern[module] > lora_alpha > rank of the down matrix (:218-238)"""
    pattern = (meta or {}).get("alpha_pattern", {}) or {}
    if module in pattern:
        return _number(pattern[module])
    base = (meta or {}).get("lora_alpha")
    if base is not None:
        try:
            return _number(base)
        except Exception:
            return None
    shape = getattr(down_weight, "shape", ())
    return float(shape[0]) if len(shape) > 0 else None


def convert_diffusers_to_comfyui(state_dict: Dict[str, Any], *, diffusion_prefix: str = "diffusion_model", adapter_metadata: Optional[dict] = None,
                                 preserve_component_prefixes: Optional[Set[str]] = None) -> Dict[str, Any]:
    """diffusers / peft keys -> `diffusion_model.<module>.lora_A|lora_B.weight` + one fp32 `.alpha` tensor per module (:333-375); component
    prefixes listed in `preserve_component_prefixes` stay (Flux / SD3 / PixArt keep `transformer.`)"""
    keep = preserve_component_prefixes or set()
    out, alphas = {}, {}
    for key, val in state_dict.items():
        for comp in COMPONENTS:
            if key.startswith(comp):
                if comp[:-1] not in keep:
                    key = f"{diffusion_prefix}." + key[len(comp):]
                break
        if ".lora.down." in key:
            key = key.replace(".lora.down.", ".lora_A.")
        elif ".lora.up." in key:
            key = key.replace(".lora.up.", ".lora_B.")
        if ".lora_A." in key:
            module = key[:key.rfind(".lora_A.")]
            a = _alpha_for(module[len(diffusion_prefix) + 1:] if module.startswith(f"{diffusion_prefix}.") else module, val, adapter_metadata)
            if a is not None and module not in alphas:
                alphas[module] = torch.tensor(a, dtype=torch.float32)
        out[key] = val
    for module, a in alphas.items():
        out[f"{module}.alpha"] = a
    return out


def convert_diffusers_to_comfyui_sd_lora(state_dict: Dict[str, Any], *, adapter_metadata: Optional[dict] = None,
                                         component_adapter_metadata: Optional[Dict[str, dict]] = None, sdxl: bool = True) -> Dict[str, Any]:
    """SD / SDXL: kohya names (`lora_unet_down_blocks_0_...`, `lora_te1_...`) that ComfyUI maps for UNet and CLIP adapters (:273-330)"""
    rename = {".lora.down.weight": ".lora_down.weight", ".lora.up.weight": ".lora_up.weight", ".lora_A.weight": ".lora_down.weight",
              ".lora_B.weight": ".lora_up.weight"}
    out, alphas = {}, {}
    for key, val in state_dict.items():
        comp = next((c for c in ("unet.", "text_encoder.", "text_encoder_2.") if key.startswith(c)), None)
        hit = _split_suffix(key, rename) if comp is not None else None
        name = comp[:-1] if comp else None
        kohya_comp = (("lora_te1" if sdxl else "lora_te") if name == "text_encoder" else KOHYA_COMPONENT.get(name)) if name else None
        if hit is None or kohya_comp is None:
            out[key] = val
            continue
        module, suf = hit
        path = module[len(comp):].replace(".processor.", ".")
        kohya = f"{kohya_comp}_{path.replace('.', '_')}"
        out[kohya + rename[suf]] = val
        if rename[suf] == ".lora_down.weight" and kohya not in alphas:
            meta = component_adapter_metadata[name] if component_adapter_metadata and name in component_adapter_metadata else adapter_metadata
            a = _alpha_for(module[len(comp):], val, meta)
            if a is not None:
                alphas[kohya] = torch.tensor(a, dtype=torch.float32)
    for kohya, a in alphas.items():
        out[f"{kohya}.alpha"] = a
    return out


def to_peft_keys(state_dict: Dict[str, Any]) -> Dict[str, Any]:
    """old-diffusers `.lora.down/.lora.up` spellings -> the peft `.lora_A/.lora_B` spellings this repo's adapters are registered under"""
    out = {}
    for key, val in state_dict.items():
        if key.endswith(".lora.down.weight"):
            key = key[:-len(".lora.down.weight")] + ".lora_A.weight"
        elif key.endswith(".lora.up.weight"):
            key = key[:-len(".lora.up.weight")] + ".lora_B.weight"
        out[key] = val
    return out
