"""The drop-in boundary as a CONTRACT: every member the reference `Trainer` touches on the model plugin, the
optimizer, the EMA object and the lr scheduler must exist on the st355 objects — names are taken from the reference source itself by AST scan, so
a reference call site the mirror forgot fails here, on the CPU, instead of as an AttributeError deep inside a training run on the GPU box.
`simpletuner_amd.integration.register()` is exercised against stand-in `simpletuner.*` modules that carry the reference's OWN registry source
(executed from where it lies; SimpleTuner itself cannot be imported here: python 3.10, no diffusers / peft — SURVEY.md F3)."""
import ast
import importlib
import sys
import types
from pathlib import Path
from types import SimpleNamespace

import pytest
import torch

REF = Path("/root/reference/simpletuner/helpers")
pytestmark = pytest.mark.skipif(not REF.exists(), reason="the reference tree is only present in the build container")


def _attr_uses(path: Path, owner: str):
    """{name: [lines]} of every `self.<owner>.<name>` in the file"""
    names = {}
    for node in ast.walk(ast.parse(path.read_text())):
        if (isinstance(node, ast.Attribute) and isinstance(node.value, ast.Attribute) and isinstance(node.value.value, ast.Name)
                and node.value.value.id == "self" and node.value.attr == owner):
            names.setdefault(node.attr, []).append(node.lineno)
    return names


def _plugins():
    from simpletuner_amd.training.trainer import default_config
    from simpletuner_amd import integration
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    out = {}
    for fam in integration.FAMILIES:
        cls = integration.plugin_class(fam)
        out[fam] = cls(default_config(model_family=fam), acc)
    return out


def test_every_model_member_the_reference_trainer_uses_exists_on_the_plugins():
    uses = _attr_uses(REF / "training" / "trainer.py", "model")
    assert len(uses) >= 50 and "prepare_batch" in uses and "check_user_config" in uses          # the scan sees the file it is meant to see
    for fam, plug in _plugins().items():
        missing = {n: l[:3] for n, l in uses.items() if not hasattr(plug, n)}
        assert not missing, f"{fam}: reference trainer.py uses self.model.<name> the st355 plugin lacks: {missing}"


def test_lifecycle_hooks_behave_like_the_reference_defaults():
    plug = _plugins()["flux"]
    # trainer.py:329-330 / 2618-2621 / 2944 / 4318 / 4376 / 4450 / 4651 / 6042 in call order: none may raise on a default config
    plug.check_user_config(); plug.validate_mixflow_config(); plug.load_text_encoder(move_to_device=False); plug.freeze_components()
    plug.pre_ema_creation(); plug.post_ema_creation(); plug.post_quantization_setup(); plug.before_accelerator_prepare()
    plug.refresh_representation_alignment_projectors(); plug.unload_text_encoder(); plug.unload_vae(); plug.apply_diffusion_blocks_trainable_filter()
    plug.diffusion_blocks_init()
    assert plug.get_text_encoder(0) is None and plug.get_text_encoder(1) is None            # trainer.py:7631-7642: no text-encoder adapters to save
    assert plug.supports_grounding() is False and plug.text_encoders == [] and plug.tokenizers == [] and plug.vae is None
    assert plug._ramtorch_base_deferred_until_after_quantization() is False
    plug.config.mixflow_enabled, plug.config.flux_fast_schedule = True, True                # common.py:4923-4950
    with pytest.raises(ValueError, match="mixflow_enabled cannot be combined with flux_fast_schedule"):
        plug.validate_mixflow_config()
    with pytest.raises(ValueError, match="configure the routes"):                             # sd3/model.py:329-337: TREAD without routes is a configuration error
        plug.tread_init()
    plug.config.tread_config = {"routes": [{"selection_ratio": 0.5, "start_layer_idx": 1, "end_layer_idx": -2}]}
    for refused in (plug.tread_init, plug.get_pipeline, plug.configure_group_offload):       # out-of-path features fail loudly, never silently (TREAD: built for SD3 and Flux; without a loaded component it refuses)
        with pytest.raises(NotImplementedError):
            refused()
    plug.unload()
    assert plug.model is None and plug.pipelines == {}


def test_optimizer_ema_and_scheduler_members_the_reference_trainer_uses_exist():
    from simpletuner_amd.training.ema import EMAModel
    from simpletuner_amd.training.optimizer import St355AdamW, St355AdamWBF16
    tr = REF / "training" / "trainer.py"
    opt_uses = set(_attr_uses(tr, "optimizer")) - {"optimizer", "optimizer_accumulation", "train", "eval"}    # accelerate-wrapper / schedule-free members (guarded by hasattr / is_schedulefree)
    assert {"step", "zero_grad", "param_groups"} <= opt_uses
    p = torch.nn.Parameter(torch.zeros(8))
    for cls in (St355AdamW, St355AdamWBF16):
        o = cls([p], lr=1e-3)
        assert not [n for n in opt_uses | {"state_dict", "load_state_dict"} if not hasattr(o, n)], cls
    ema_uses = set(_attr_uses(tr, "ema_model"))
    assert {"step", "copy_to", "to"} <= ema_uses
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    ema = EMAModel(SimpleNamespace(ema_device="accelerator", ema_cpu_only=False, ema_update_interval=None), acc, [p], decay=0.99)
    assert not [n for n in ema_uses | {"store", "restore", "state_dict", "save_pretrained"} if not hasattr(ema, n)]
    sched_uses = set(_attr_uses(tr, "lr_scheduler")) - {"num_update_steps_per_epoch"}       # guarded by hasattr (trainer.py:3895)
    from simpletuner_amd.training.lr_schedule import get_lr_scheduler
    cfg = SimpleNamespace(lr_scheduler="cosine", learning_rate=1e-3, lr_end=1e-5, lr_warmup_steps=2, max_train_steps=10, lr_num_cycles=1, lr_power=1.0,
                          num_update_steps_per_epoch=5, gradient_accumulation_steps=1, is_schedulefree=False, use_deepspeed_scheduler=False)
    sch = get_lr_scheduler(cfg, St355AdamW([p], lr=1e-3), acc, None, 0)
    assert not [n for n in sched_uses if not hasattr(sch, n)]


def _stand_in_simpletuner(monkeypatch):
    """`simpletuner.helpers.models.registry` = the reference's own registry.py executed in place; `common.ModelFoundation` and
    `optimizer_param.optimizer_choices` as minimal stand-ins with the reference's shapes (their real modules import diffusers)"""
    mods = {}
    for name in ("simpletuner", "simpletuner.helpers", "simpletuner.helpers.models", "simpletuner.helpers.training"):
        mods[name] = types.ModuleType(name)
        mods[name].__path__ = []
    reg = types.ModuleType("simpletuner.helpers.models.registry")
    reg.__file__ = str(REF / "models" / "registry.py")
    exec(compile((REF / "models" / "registry.py").read_text(), reg.__file__, "exec"), reg.__dict__)
    common = types.ModuleType("simpletuner.helpers.models.common")

    import abc
    import ast as _ast
    tree = _ast.parse((REF / "models" / "common.py").read_text())
    ref_cls = next(n for n in tree.body if isinstance(n, _ast.ClassDef) and n.name == "ModelFoundation")
    abstract = [f.name for f in ref_cls.body if isinstance(f, _ast.FunctionDef)
                and any((isinstance(d, _ast.Name) and d.id == "abstractmethod") or (isinstance(d, _ast.Attribute) and d.attr == "abstractmethod") for d in f.decorator_list)]
    assert {"model_predict", "_encode_prompts", "convert_text_embed_for_pipeline", "convert_negative_text_embed_for_pipeline"} <= set(abstract), abstract

    def _abstract(name):
        def f(self, *a, **k):
            raise NotImplementedError(name)
        f.__name__ = name
        return abc.abstractmethod(f)

    ns = {n: _abstract(n) for n in abstract}         # the reference base IS an ABC (common.py:443): every abstract member it declares, by name
    ns.update(REFERENCE_BASE=True, reference_only_helper=lambda self: "from the reference base")
    ModelFoundation = abc.ABCMeta("ModelFoundation", (abc.ABC,), ns)
    common.ModelFoundation = ModelFoundation
    common.ABSTRACT = abstract
    optp = types.ModuleType("simpletuner.helpers.training.optimizer_param")
    optp.optimizer_choices = {"torch-adamw": {"precision": "any", "default_settings": {"betas": (0.9, 0.999), "weight_decay": 1e-2, "eps": 1e-8},
                                              "class": torch.optim.AdamW},
                              "adamw_bf16": {"precision": "bf16", "default_settings": {"betas": (0.9, 0.999), "weight_decay": 1e-2, "eps": 1e-6},
                                             "class": object, "gradient_precision": "fp32"}}
    mods.update({reg.__name__: reg, common.__name__: common, optp.__name__: optp})
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    return reg, common, optp


def test_register_into_the_reference_registry_and_optimizer_choices(monkeypatch):
    from simpletuner_amd import integration
    from simpletuner_amd.training.optimizer import St355AdamW, St355AdamWBF16
    reg, common, optp = _stand_in_simpletuner(monkeypatch)
    out = integration.register()
    fams = reg.ModelRegistry.model_families()                      # the reference's own lookup: explicit registrations override lazy metadata entries
    for fam in integration.FAMILIES:
        cls = fams[fam]
        assert cls is out[fam] and issubclass(cls, common.ModelFoundation) and getattr(cls, "ST355_NATIVE", False)
        assert cls.__mro__[1].__module__.startswith("simpletuner_amd.")            # st355 implementation first, reference base behind it
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    from simpletuner_amd.training.trainer import default_config
    plug = fams["flux"](default_config(model_family="flux"), acc)                  # trainer.py:329: ModelRegistry.model_families()[family](config, accelerator)
    assert isinstance(plug, common.ModelFoundation) and plug.reference_only_helper() == "from the reference base"
    for fam in integration.FAMILIES:                                                # the reference base is an ABC: every registered class must be CONSTRUCTIBLE
        assert not getattr(fams[fam], "__abstractmethods__", None), (fam, fams[fam].__abstractmethods__)
        inst = fams[fam](default_config(model_family=fam), acc)
        rec = {"prompt_embeds": torch.zeros(5, 8), "pooled_prompt_embeds": torch.zeros(6), "attention_mask": torch.ones(5), "attention_masks": torch.ones(5)}
        pos, neg = inst.convert_text_embed_for_pipeline(rec), inst.convert_negative_text_embed_for_pipeline(rec)
        assert pos["prompt_embeds"].shape == (1, 5, 8)
        assert fam == "flux" or neg["negative_prompt_embeds"].shape == (1, 5, 8)           # Flux: {} unless real CFG is configured (flux/model.py:476-478)
        with pytest.raises(NotImplementedError, match="text encoders are not part of the st355 per-step path"):
            inst._encode_prompts(["a prompt"])
    assert fams["pixart_sigma"](default_config(model_family="pixart_sigma"), acc).convert_text_embed_for_pipeline(
        {"prompt_embeds": torch.zeros(5, 8), "attention_mask": torch.ones(5)})["prompt_attention_mask"].shape == (1, 5)
    assert type(plug).prepare_batch.__module__ == "simpletuner_amd.foundation"      # the step path is the MI355X one
    ch = optp.optimizer_choices
    assert ch["st355-adamw"]["class"] is St355AdamW and ch["adamw_bf16"]["class"] is St355AdamWBF16
    assert ch["adamw_bf16"]["gradient_precision"] == "fp32" and ch["adamw_bf16"]["precision"] == "bf16"      # untouched keys of the entry survive
    assert ch["torch-adamw"]["class"] is torch.optim.AdamW
    for e in (ch["st355-adamw"], ch["adamw_bf16"]):
        assert {"precision", "default_settings", "class"} <= set(e)
    assert integration.ema_class().__name__ == "EMAModel"


def test_register_without_simpletuner_fails_loudly(monkeypatch):
    from simpletuner_amd import integration
    for k in [k for k in sys.modules if k == "simpletuner" or k.startswith("simpletuner.")]:
        monkeypatch.delitem(sys.modules, k)
    monkeypatch.setattr(importlib, "import_module", lambda name, *a, **k: (_ for _ in ()).throw(ImportError(f"No module named {name!r}"))
                        if name.startswith("simpletuner.") else importlib.__import__(name))
    with pytest.raises(integration.IntegrationUnavailable, match="needs an importable SimpleTuner"):
        integration.register()


def test_reference_family_members_that_reach_into_the_component_are_guarded(monkeypatch, tmp_path):
    """`plugin_class(family, ModelFoundation, <the reference's own family class>)`: helpers of the reference FAMILY class that touch `self.model` as a diffusers module
    and have no st355 implementation refuse loudly instead of silently running against the st355 component; members that do not touch it, members the st355 class
    defines, and the allow-listed ones keep resolving as before"""
    import importlib.util
    from simpletuner_amd import integration
    _, common, _ = _stand_in_simpletuner(monkeypatch)
    src = tmp_path / "fake_ref_flux.py"
    src.write_text(
        "from simpletuner.helpers.models.common import ModelFoundation\n"
        "class Flux(ModelFoundation):\n"
        "    def control_init(self):\n"
        "        return self.unwrap_model(self.model).x_embedder.weight\n"
        "    def _maybe_load_assistant_lora(self):\n"
        "        return self.get_trained_component().load_lora_adapter('x')\n"
        "    def custom_model_card_schedule_info(self):\n"
        "        return [str(self.model)]\n"
        "    def harmless_helper(self):\n"
        "        return 'no component involved'\n"
        "    def model_predict(self, batch):\n"
        "        return self.model(**batch)\n")
    spec = importlib.util.spec_from_file_location("fake_ref_flux", src)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cls = integration.plugin_class("flux", common.ModelFoundation, mod.Flux)
    assert issubclass(cls, mod.Flux) and cls.__mro__[1].__module__.startswith("simpletuner_amd.")
    acc = SimpleNamespace(device=torch.device("cpu"), num_processes=1, process_index=0, is_main_process=True)
    from simpletuner_amd.training.trainer import default_config
    inst = cls(default_config(model_family="flux"), acc)
    for name in ("control_init", "_maybe_load_assistant_lora"):
        with pytest.raises(NotImplementedError, match=f"St355Flux.{name}: this member of the reference's Flux"):
            getattr(inst, name)()
    assert inst.harmless_helper() == "no component involved"                       # does not touch the component: reference code, as before
    assert cls.custom_model_card_schedule_info is mod.Flux.custom_model_card_schedule_info          # allow-listed
    assert cls.model_predict.__module__.startswith("simpletuner_amd.")             # the st355 class defines it: the MRO picks the MI355X implementation
