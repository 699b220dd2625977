"""Binding the MI355X hot path INTO an installed SimpleTuner.

Where `simpletuner` imports (Python >= 3.12 + diffusers + peft, none of which exist in the build container — SURVEY.md F3), one call
wires this package behind the reference's own plugin surface:

    import simpletuner_amd.integration as st355
    st355.register()          # before `Trainer(...)` / `train.py` builds the model

  * every st355 family plugin is re-created as a class that ALSO derives from the reference's `ModelFoundation`
    (simpletuner/helpers/models/common.py:451) — so `isinstance` checks in the trainer hold — with the st355 class first in the MRO, and
    is registered with `ModelRegistry.register(family, cls)` (simpletuner/helpers/models/registry.py:74-75), overriding the lazy
    metadata entry of the same family (`model_families()` lets explicit registrations win, helpers/models/registry.py:87-98);
  * `optimizer_choices` (simpletuner/helpers/training/optimizer_param.py:76) gains `st355-adamw` and has its `adamw_bf16` entry's class
    replaced by the fused one-pass implementation (same entry shape: precision / default_settings / class);
  * `EMAModel` is exported for `trainer.py:4360-4369`'s construction site (same constructor signature, ema.py:29-60).

`tests/test_integration_contract_cpu.py` holds this module to the reference: it AST-scans /root/reference's trainer.py for every
`self.model.<member>` and optimizer / EMA member the Trainer uses and fails on any name the plugin objects lack, and exercises
`register()` against stand-in `simpletuner` modules with the reference's own registry source.
"""
from __future__ import annotations

import importlib
from typing import Dict, Optional

FAMILIES = {
    "flux": ("simpletuner_amd.flux.model", "Flux"),
    "sd3": ("simpletuner_amd.sd3.model", "SD3"),
    "sdxl": ("simpletuner_amd.sdxl.model", "SDXL"),
    "sd1x": ("simpletuner_amd.sd1x.model", "StableDiffusion1"),
    "pixart_sigma": ("simpletuner_amd.pixart.model", "PixartSigma"),
}


class IntegrationUnavailable(ImportError):
    pass


def _ref(module: str):
    try:
        return importlib.import_module(module)
    except Exception as e:          # ImportError, or a syntax / dependency failure deep inside the reference
        raise IntegrationUnavailable(f"cannot import {module}: {e}.  simpletuner_amd.integration.register() needs an importable SimpleTuner "
                                     f"(python >= 3.12 with diffusers / peft); the st355 host harness (simpletuner_amd.training.trainer) "
                                     f"replays the same step order where it is not installed") from e


def plugin_class(family: str, ref_foundation: Optional[type] = None, ref_family: Optional[type] = None) -> type:
    """the st355 plugin of `family`; with `ref_foundation` a subclass that also derives from the reference's ModelFoundation — through the
    reference's OWN family class (`ref_family`, e.g. simpletuner.helpers.models.flux.model.Flux) when that is importable, so that everything off the
    step path (text encoders / `_encode_prompts`, pipelines, checkpoint loading) keeps resolving to the reference's implementation"""
    mod, name = FAMILIES[family]
    cls = getattr(importlib.import_module(mod), name)
    if ref_foundation is None or issubclass(cls, ref_foundation):
        return cls
    base = ref_family if (isinstance(ref_family, type) and issubclass(ref_family, ref_foundation)) else ref_foundation
    # st355 first in the MRO: every step-path method resolves to the MI355X implementation; the reference class behind it contributes identity
    # (isinstance) and whatever out-of-path member the st355 class does not define
    ns = {"__module__": __name__, "__doc__": cls.__doc__, "ST355_NATIVE": True}
    ns.update(_guards_for_reference_members(cls, base, ref_foundation, name))
    new = type(f"St355{name}", (cls, base), ns)
    left = sorted(getattr(new, "__abstractmethods__", ()))
    if left:           # an ABC base with members neither side implements: the class could be registered but never constructed — fail at register() time
        raise TypeError(f"St355{name} would be abstract: {left} are declared abstract by {base.__name__} and implemented by neither class")
    return new


# members of a reference FAMILY class that may keep resolving to reference code although they mention the trained component: they only read its `config`, hand it
# to the reference's own pipeline / text-encoder code, or are hooks the st355 foundation answers through a differently named method
REFERENCE_MEMBERS_OK = frozenset({
    "_format_text_embedding", "_encode_prompts", "convert_text_embed_for_pipeline", "convert_negative_text_embed_for_pipeline", "update_pipeline_call_kwargs",
    "pretrained_load_args", "get_pipeline", "_load_pipeline", "setup_model_flavour", "custom_model_card_schedule_info", "custom_model_card_code_example",
    # helpers that only read the component's `config` (PixArt: the kwargs its reference `_load_pipeline` builds, the latent sequence length)
    "_pixart_from_pretrained_kwargs", "_latent_sequence_length",
})


def _guards_for_reference_members(cls: type, base: type, ref_foundation: type, name: str) -> dict:
    """A family class of the reference defines helpers next to its step path that reach into `self.model` as a diffusers module (`_prepare_model_predict_timesteps`,
    `control_init`, `post_model_load_setup`, `_maybe_load_assistant_lora`, …).  With the reference family class behind the st355 class in the MRO, such a member the
    st355 class does not define would silently run reference code against the st355 component.  Every function that the reference FAMILY class itself adds (not its
    ModelFoundation), that the st355 class does not define, whose source touches the trained component, and that is not in REFERENCE_MEMBERS_OK is replaced by a
    stub that refuses loudly, naming itself.  (Source not retrievable -> the member is left alone.)"""
    import inspect
    out = {}
    if base is ref_foundation:
        return out
    for klass in base.__mro__:
        if klass is ref_foundation or klass is object or issubclass(ref_foundation, klass):
            continue                                  # ModelFoundation and its own bases: the contract the st355 foundation mirrors (test_integration_contract_cpu.py)
        for attr, val in vars(klass).items():
            if attr.startswith("__") or attr in out or attr in REFERENCE_MEMBERS_OK or not inspect.isfunction(val):
                continue
            if any(attr in vars(k) for k in cls.__mro__ if k is not object):
                continue                              # the st355 side defines it: its implementation wins the MRO anyway
            try:
                src = inspect.getsource(val)
            except (OSError, TypeError):
                continue
            if not any(tok in src for tok in ("self.model", "get_trained_component", "unwrap_model", "self.controlnet")):
                continue

            def refuse(self, *a, _attr=attr, _klass=klass.__name__, **k):
                raise NotImplementedError(f"St355{name}.{_attr}: this member of the reference's {_klass} reaches into the trained component as a diffusers module and "
                                          f"has no st355 implementation (simpletuner_amd.integration.REFERENCE_MEMBERS_OK lists the members that may resolve to reference code)")
            refuse.__name__, refuse.__qualname__ = attr, f"St355{name}.{attr}"
            out[attr] = refuse
    return out


def _reference_family_class(reg, family: str) -> Optional[type]:
    """the reference's own class of `family`, if its module imports here (registry.py:77-90: explicit registration, else the lazy metadata entry)"""
    try:
        got = reg.get(family)
        if got is None:
            return None
        if hasattr(got, "get_real_class"):
            got = got.get_real_class()
        return got if isinstance(got, type) and not getattr(got, "ST355_NATIVE", False) else None
    except Exception:
        return None


def register(families=None, overwrite_optimizers: bool = True) -> Dict[str, type]:
    """register the st355 plugins / optimizers into the installed SimpleTuner; returns {family: registered class}"""
    reg = _ref("simpletuner.helpers.models.registry").ModelRegistry
    ref_foundation = getattr(_ref("simpletuner.helpers.models.common"), "ModelFoundation")
    out = {}
    for fam in (families or FAMILIES):
        cls = plugin_class(fam, ref_foundation, _reference_family_class(reg, fam))
        reg.register(fam, cls)
        out[fam] = cls
    register_optimizers(overwrite=overwrite_optimizers)
    return out


def register_optimizers(overwrite: bool = True) -> Dict[str, dict]:
    """add / replace the entries of the reference's `optimizer_choices` (same dict shape as optimizer_param.py:76-96)"""
    from .training.optimizer import OPTIMIZER_CHOICE
    choices = _ref("simpletuner.helpers.training.optimizer_param").optimizer_choices
    for name, entry in OPTIMIZER_CHOICE.items():
        if name in choices and not overwrite:
            continue
        merged = dict(choices.get(name, {}))
        merged.update({"precision": entry["precision"], "default_settings": dict(entry["default_settings"]), "class": entry["class"]})
        choices[name] = merged
    return choices


def ema_class():
    from .training.ema import EMAModel
    return EMAModel
