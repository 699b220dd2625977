#!/usr/bin/env python3
"""Generate tests/golden/*.pt by EXECUTING THE REFERENCE'S OWN CODE in this container.

The reference package cannot be imported here (python 3.10 < required 3.12; diffusers/peft absent), so the
self-contained, pure-torch functions on the hot path are pulled out of the reference source files BY AST at generation
time (never copied into this repo), compiled in a namespace that provides only `torch`/`math`, and run on seeded inputs.
The resulting tensors are committed as small fixtures; tests/test_golden_cpu.py pins the oracle to them and
tests/test_golden_gpu.py pins the HIP kernels to them.  /root/reference is read ONLY by this script, never at test time.

    python tools/gen_golden.py            (writes tests/golden/reference_vectors.pt)
"""
from __future__ import annotations

import ast
import math
import sys
import types
from pathlib import Path

import torch

REF = Path("/root/reference/simpletuner")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "reference_vectors.pt"


def extract(path: Path, names, extra_ns=None, class_name=None):
    """compile the named top-level functions (or methods of `class_name`) of a reference file, in isolation"""
    src = path.read_text()
    tree = ast.parse(src)
    body = tree.body
    if class_name is not None:
        cls = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == class_name)
        body = cls.body
    picked = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise KeyError(f"{path}: missing {missing}")
    for n in picked:
        n.decorator_list = []
    mod = ast.Module(body=picked, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"torch": torch, "math": math, "Optional": None, "__builtins__": __builtins__}
    ns.update(extra_ns or {})
    exec(compile(mod, str(path), "exec"), ns)
    return [ns[n] for n in names]


def gen_adamw_bf16():
    """AdamWBF16 (the examples' default optimizer): run the reference CLASS ITSELF (optimizers/adamw_bfloat16/__init__.py:20-180 +
    stochastic/__init__.py:47-124, imported from /root/reference by file path — pure torch) for 4 steps on two bf16 tensors, recording
    every random draw (`torch.randint_like` of copy_stochastic_, `torch.rand` of the decay phase) so the oracle / the HIP kernel can be
    fed the SAME stochastic-rounding bits and must reproduce the states bit for bit."""
    import importlib.util

    pkg_dir = REF / "helpers/training/optimizers/adamw_bfloat16"
    spec = importlib.util.spec_from_file_location("ref_adamw_bf16", pkg_dir / "__init__.py", submodule_search_locations=[str(pkg_dir)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_adamw_bf16"] = mod
    spec.loader.exec_module(mod)

    torch.manual_seed(4321)
    draws, rands = [], []
    real_randint_like, real_rand = torch.randint_like, torch.rand

    def rec_randint_like(*a, **k):
        r = real_randint_like(*a, **k)
        draws.append(r.clone())
        return r

    def rec_rand(*a, **k):
        r = real_rand(*a, **k)
        rands.append(float(r))
        return r

    shapes = [(37, 29), (515,)]
    params = [torch.nn.Parameter((torch.randn(s) * 0.5).to(torch.bfloat16)) for s in shapes]
    lr, wd, betas, eps = 1e-2, 0.2, (0.9, 0.999), 1e-8
    opt = mod.AdamWBF16(params, lr=lr, betas=betas, eps=eps, weight_decay=wd)
    G = {"p0": [p.detach().clone() for p in params], "lr": lr, "wd": wd, "betas": betas, "eps": eps, "steps": []}
    torch.randint_like, torch.rand = rec_randint_like, rec_rand
    try:
        for step in range(4):
            grads = [(torch.randn(s) * (0.1 + 0.3 * step)).to(torch.bfloat16) for s in shapes]
            for p, g in zip(params, grads):
                p.grad = g.clone()
            draws.clear()
            opt.step()
            assert len(draws) == 4 * len(params), len(draws)     # exp_avg, shift(addcdiv), p, shift(error) per parameter
            G["steps"].append({
                "grads": grads,
                "draws": [[d.clone() for d in draws[4 * i:4 * i + 4]] for i in range(len(params))],
                "p": [p.detach().clone() for p in params],
                "exp_avg": [opt.state[p]["exp_avg"].clone() for p in params],
                "exp_avg_sq": [opt.state[p]["exp_avg_sq"].clone() for p in params],
                "shift": [opt.state[p]["shift"].clone() for p in params],
                "accumulated_decay": [float(opt.state[p]["accumulated_decay"]) for p in params],
            })
    finally:
        torch.randint_like, torch.rand = real_randint_like, real_rand
    G["accumulated_decay0"] = list(rands)      # torch.rand([]) * decay_threshold is the initial value (lazy state init, first step)
    G["decay_threshold"] = float(mod.AdamWBF16.decay_threshold)
    G["_cite"] = "simpletuner/helpers/training/optimizers/adamw_bfloat16/__init__.py:55-180; stochastic/__init__.py:47-124"
    out = OUT.parent / "adamw_bf16_vectors.pt"
    torch.save(G, out)
    print(f"wrote {out}: {len(G['steps'])} steps x {len(shapes)} tensors; initial decay draws {rands}")


def gen_loss():
    """conditional_loss / compute_scheduled_huber_c (common.py:6132-6216) and compute_snr (min_snr_gamma.py:4-41): the reference METHODS
    lifted by AST and run on seeded inputs -> tests/golden/loss_vectors.pt (oracle + HIP kernel are pinned to these)."""
    import enum
    import torch.nn.functional as F

    class PredictionTypes(enum.Enum):
        EPSILON = "epsilon"; V_PREDICTION = "v_prediction"; FLOW_MATCHING = "flow_matching"

    (compute_snr,) = extract(REF / "helpers/training/min_snr_gamma.py", ["compute_snr"])
    cond_loss, sched_c = extract(REF / "helpers/models/common.py", ["conditional_loss", "compute_scheduled_huber_c"], class_name="ModelFoundation",
                                 extra_ns={"F": F, "PredictionTypes": PredictionTypes, "compute_snr": compute_snr})
    torch.manual_seed(777)
    G = {}
    pred = torch.randn(3, 16, 8, 8); target = torch.randn(3, 16, 8, 8)
    G["pred"], G["target"] = pred.to(torch.bfloat16), target.to(torch.bfloat16)
    pf, tf = G["pred"].float(), G["target"].float()            # the reference calls conditional_loss on .float() tensors (common.py:6275-6281)
    for lt in ("l2", "huber", "smooth_l1"):
        for c in (0.1, 0.03):
            el = cond_loss(None, pf, tf, reduction="none", loss_type=lt, huber_c=c)
            G[f"{lt}.c{c}.per_sample"] = el.mean(dim=[1, 2, 3])
            G[f"{lt}.c{c}.loss"] = el.mean(dim=[1, 2, 3]).mean()                      # common.py:6426-6429
    ts = torch.tensor([10.0, 250.0, 500.0, 900.0, 999.0])
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2      # DDPM "scaled_linear" (SD1.5 / SDXL scheduler_config)
    acp = torch.cumprod(1.0 - betas, dim=0)
    sched = types.SimpleNamespace(alphas_cumprod=acp, config=types.SimpleNamespace(num_train_timesteps=1000))
    G["timesteps"] = ts; G["alphas_cumprod"] = acp
    for schedule in ("constant", "exponential", "snr"):
        for ptype in (PredictionTypes.FLOW_MATCHING, PredictionTypes.EPSILON):
            self_ = types.SimpleNamespace(config=types.SimpleNamespace(loss_type="huber", huber_schedule=schedule, huber_c=0.1),
                                          noise_schedule=sched, PREDICTION_TYPE=ptype)
            t_in = ts if ptype == PredictionTypes.FLOW_MATCHING else ts.long()
            G[f"huber_c.{schedule}.{ptype.value}"] = torch.as_tensor(sched_c(self_, t_in)).float()
    tl = torch.tensor([0, 1, 250, 500, 998, 999])
    G["snr.t"] = tl
    G["snr"] = compute_snr(tl, sched)
    G["snr.soft_min"] = compute_snr(tl, sched, use_soft_min=True, sigma_data=1.0)
    G["_cite"] = "simpletuner/helpers/models/common.py:6132-6216; simpletuner/helpers/training/min_snr_gamma.py:4-41"
    out = OUT.parent / "loss_vectors.pt"
    torch.save(G, out)
    print(f"wrote {out}: {len(G)} entries")


def gen_fp8():
    """fp8-native Linear (quantisation/fp8_native.py:25-119): the reference MODULE imported by file path (pure torch).  Weight quantisation is
    its own function; for the forward, torch._scaled_mm is replaced by a recorder so that the reference's own code produces the e5m2
    activations and the scale vectors it would hand to the device GEMM (the CPU backend of _scaled_mm has no row-wise mode)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_fp8_native", REF / "helpers/training/quantisation/fp8_native.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_fp8_native"] = mod
    spec.loader.exec_module(mod)
    torch.manual_seed(2468)
    G = {}
    w = (torch.randn(264, 384) * 0.05).to(torch.bfloat16)
    w[5] = 0                                            # an all-zero row exercises the amax clamp
    bias = torch.randn(264).to(torch.bfloat16)
    x = (torch.randn(3, 50, 384) * 1.7).to(torch.bfloat16)
    q, sc = mod.quantize_weight_to_fp8(w)
    G["w"], G["bias"], G["x"] = w, bias, x
    G["w_q"], G["w_scale"] = q.view(torch.uint8), sc
    rec = {}
    real = torch._scaled_mm

    def recorder(a, b, scale_a=None, scale_b=None, bias=None, out_dtype=None, use_fast_accum=False):
        rec.update(x_q=a.clone().view(torch.uint8), scale_a=scale_a.clone(), scale_b=scale_b.clone(), b_is_wT=bool(torch.equal(b.t().view(torch.uint8), q.view(torch.uint8))))
        out = (a.float() @ b.float()) * scale_a * scale_b              # row-wise scaling semantics of _scaled_mm, fp32
        if bias is not None:
            out = out + bias.float()
        return out.to(out_dtype)

    mod._scaled_mm_supported = lambda t: True
    torch._scaled_mm = recorder
    try:
        out = mod._Fp8NativeLinearFn.apply(x, q, sc, bias, 264)
    finally:
        torch._scaled_mm = real
    assert rec["b_is_wT"]
    G["x_q"], G["scale_a"], G["scale_b"] = rec["x_q"], rec["scale_a"], rec["scale_b"]
    G["out_fp32_semantics"] = out                           # reference control flow + fp32 emulation of the device GEMM
    go = torch.randn(3, 50, 264).to(torch.bfloat16)
    xg = x.clone().requires_grad_(True)
    torch._scaled_mm = recorder
    try:
        mod._Fp8NativeLinearFn.apply(xg, q, sc, bias, 264).backward(go)
    finally:
        torch._scaled_mm = real
    G["grad_out"], G["grad_x"] = go, xg.grad                # backward = dequantised-weight matmul (fp8_native.py:107-115)
    G["_cite"] = "simpletuner/helpers/training/quantisation/fp8_native.py:25-119"
    out_p = OUT.parent / "fp8_vectors.pt"
    torch.save(G, out_p)
    print(f"wrote {out_p}: scale_a {float(rec['scale_a'][0])}")


def gen_ddpm_sampling():
    """generate_timestep_weights + segmented_timestep_selection (helpers/training/custom_schedule.py:18-100) executed as written, with the
    torch RNG pinned -> tests/golden/ddpm_sampling_vectors.pt"""
    from types import SimpleNamespace
    p = REF / "helpers" / "training" / "custom_schedule.py"
    seg, gen_w = extract(p, ["segmented_timestep_selection", "generate_timestep_weights"])
    G = {"weights": {}, "segmented": []}
    for strat, extra in (("none", {}), ("later", dict(timestep_bias_portion=0.25, timestep_bias_multiplier=2.0)),
                         ("earlier", dict(timestep_bias_portion=0.5, timestep_bias_multiplier=3.0)),
                         ("range", dict(timestep_bias_portion=0.25, timestep_bias_multiplier=4.0, timestep_bias_begin=200, timestep_bias_end=500))):
        args = SimpleNamespace(timestep_bias_strategy=strat, timestep_bias_portion=0.25, timestep_bias_multiplier=1.0, timestep_bias_begin=0, timestep_bias_end=1000)
        for k, v in extra.items():
            setattr(args, k, v)
        G["weights"][strat] = (dict(vars(args)), gen_w(args, 1000).clone())
    ezt, = extract(p, ["enforce_zero_terminal_snr"])
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    G["zero_terminal_snr_betas"] = ezt(betas.clone()).clone()
    cfg = SimpleNamespace(refiner_training=False, refiner_training_invert_schedule=False, refiner_training_strength=0.2)
    for bsz in (2, 4, 7):
        for seed in (0, 1, 2):
            torch.manual_seed(seed)
            G["segmented"].append((bsz, seed, seg(1000, bsz, torch.ones(1000), cfg).clone()))
    G["segmented_refiner"] = []
    for invert, strength in ((False, 0.2), (True, 0.2), (False, 0.35), (True, 0.8)):
        rcfg = SimpleNamespace(refiner_training=True, refiner_training_invert_schedule=invert, refiner_training_strength=strength)
        for bsz in (2, 5):
            torch.manual_seed(11)
            G["segmented_refiner"].append((invert, strength, bsz, 11, seg(1000, bsz, torch.ones(1000), rcfg).clone()))
    out = OUT.parent / "ddpm_sampling_vectors.pt"
    torch.save(G, out)
    print("wrote", out)


def gen_cubic_schedule():
    """CubicSplineDistribution (helpers/training/timestep_distribution.py:57-187; torch + stdlib only, so the file is loaded as a module as it
    lies): tabulated pdf / cdf, seeded samples, log_prob -> tests/golden/cubic_schedule_vectors.pt"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_timestep_distribution", REF / "helpers" / "training" / "timestep_distribution.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    G = {"cases": [], "parse": []}
    for weights in ([0.0, 1.0], [1.0, 0.0], [0.0, 1.0, 0.1, 2.0, 0.0], [1.0, 1.0, 1.0], [0.2, 3.0, 0.2], [0.0, 0.5, 4.0, 4.0, 0.5, 0.0, 1.0], [5.0, 0.0, 0.0, 5.0]):
        d = mod.CubicSplineDistribution(weights)
        torch.manual_seed(23)
        smp = d.sample((257,))
        q = torch.linspace(-0.1, 1.1, 49)
        slopes = d._pchip_slopes(torch.tensor(weights), 1.0 / (len(weights) - 1))
        G["cases"].append(dict(weights=list(weights), pdf=d.pdf_grid.clone(), cdf=d.cdf_grid.clone(), seed=23, samples=smp.clone(), query=q, log_prob=d.log_prob(q).clone(),
                               slopes=slopes.clone()))
    for raw in (None, "", "none", "[0, 1, 0.5]", "0,1,0.5", "0; 1 ;0.5", [], [4.0], 3, torch.tensor([[0.0, 2.0]]), (1, 2)):
        G["parse"].append((raw, mod.parse_cubic_spline_weights(raw)))
    out = OUT.parent / "cubic_schedule_vectors.pt"
    torch.save(G, out)
    print("wrote", out)


def gen_collate():
    """compute_time_ids + gather_conditional_sdxl_size_features (helpers/training/collate.py:59-98, 501-523) executed as written (StateTracker /
    logger stubbed: not the refiner) -> tests/golden/collate_vectors.pt"""
    from types import SimpleNamespace
    stub_log = SimpleNamespace(debug=lambda *a, **k: None)
    stub_state = SimpleNamespace(is_sdxl_refiner=lambda: False)
    p = REF / "helpers" / "training" / "collate.py"
    cti, = extract(p, ["compute_time_ids"], extra_ns={"logger": stub_log, "StateTracker": stub_state})
    gss, = extract(p, ["gather_conditional_sdxl_size_features"], extra_ns={"logger": stub_log, "StateTracker": stub_state, "compute_time_ids": cti})
    G = {"time_ids": [], "sdxl": []}
    for inter, tgt, crop in (((1024, 1024), (4, 128, 128), (0, 0)), ((1344, 896), (4, 96, 160), (64, 32)), ((640, 1536), (4, 192, 80), (0, 128))):
        for dt in (torch.float32, torch.bfloat16):
            G["time_ids"].append((inter, tgt, crop, dt, cti(inter, tgt, dt, crop_coordinates=list(crop))))
    examples = [dict(intermediary_size=(1100, 1024), crop_coordinates=(0, 38), drop_conditioning=False),
                dict(original_size=(2048, 2048), crop_coordinates=(10, 20), drop_conditioning=True),
                dict(intermediary_size=(1024, 1400), original_size=(1, 1), crop_coordinates=(188, 0), drop_conditioning=False)]
    lat = torch.zeros(3, 4, 128, 128)
    G["sdxl"] = (examples, tuple(lat.shape), gss(examples, lat, torch.bfloat16))
    out = OUT.parent / "collate_vectors.pt"
    torch.save(G, out)
    print("wrote", out)


def gen_flow_match_scheduler():
    """the flow-match Euler scheduler vendored in the reference tree (helpers/models/ace_step/schedulers/scheduling_flow_match_euler_discrete.py)
    loaded as a module with its three diffusers imports shimmed (ConfigMixin / SchedulerMixin / register_to_config / BaseOutput / logging: no
    arithmetic lives in them) -> tests/golden/flow_match_scheduler_vectors.pt"""
    import importlib.util
    import inspect
    from types import SimpleNamespace

    def register_to_config(init):
        def wrapped(self, *a, **kw):
            b = inspect.signature(init).bind(self, *a, **kw)
            b.apply_defaults()
            self.config = SimpleNamespace(**{k: v for k, v in b.arguments.items() if k != "self"})
            init(self, *a, **kw)
        return wrapped

    class _Out(dict):
        pass

    shims = {"diffusers": types.ModuleType("diffusers"), "diffusers.configuration_utils": types.ModuleType("diffusers.configuration_utils"),
             "diffusers.schedulers": types.ModuleType("diffusers.schedulers"), "diffusers.schedulers.scheduling_utils": types.ModuleType("diffusers.schedulers.scheduling_utils"),
             "diffusers.utils": types.ModuleType("diffusers.utils")}
    shims["diffusers.configuration_utils"].ConfigMixin = type("ConfigMixin", (), {})
    shims["diffusers.configuration_utils"].register_to_config = register_to_config
    shims["diffusers.schedulers.scheduling_utils"].SchedulerMixin = type("SchedulerMixin", (), {})
    shims["diffusers.utils"].BaseOutput = _Out
    shims["diffusers.utils"].logging = SimpleNamespace(get_logger=lambda *_a, **_k: SimpleNamespace(warning=print, info=print, debug=lambda *a, **k: None))
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        spec = importlib.util.spec_from_file_location("_ref_fm_sched", REF / "helpers" / "models" / "ace_step" / "schedulers" / "scheduling_flow_match_euler_discrete.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    Sch = mod.FlowMatchEulerDiscreteScheduler
    G = {"cases": []}
    g = torch.Generator().manual_seed(4)
    for kw, steps, mu in ((dict(num_train_timesteps=10, shift=3.0), 10, None), (dict(num_train_timesteps=1000, shift=3.0), 28, None),
                          (dict(num_train_timesteps=1000, shift=1.0), 20, None), (dict(num_train_timesteps=1000, use_dynamic_shifting=True), 28, 1.15),
                          (dict(num_train_timesteps=1000, use_dynamic_shifting=True), 4, 0.5)):
        sc = Sch(**kw)
        rec = dict(kw=kw, steps=steps, mu=mu, init_sigmas=sc.sigmas.clone(), init_timesteps=sc.timesteps.clone(), sigma_min=sc.sigma_min, sigma_max=sc.sigma_max)
        sc.set_timesteps(num_inference_steps=steps, mu=mu)
        rec.update(sigmas=sc.sigmas.clone(), timesteps=sc.timesteps.clone())
        x = torch.randn(2, 4, 8, 8, generator=g)
        traj = [x.clone()]
        vs = []
        for t in sc.timesteps:
            v = torch.randn(2, 4, 8, 8, generator=g)
            vs.append(v)
            x = sc.step(v, t, x, return_dict=False)[0]
            traj.append(x.clone())
        rec.update(v=torch.stack(vs), traj=torch.stack(traj))
        sc2 = Sch(**kw)
        sc2.set_timesteps(num_inference_steps=steps, mu=mu)
        smp, noi = torch.randn(3, 4, 8, 8, generator=g), torch.randn(3, 4, 8, 8, generator=g)
        ts = sc2.timesteps[[0, steps // 2, steps - 1]]
        rec.update(sn_sample=smp, sn_noise=noi, sn_t=ts.clone(), sn_out=sc2.scale_noise(smp, ts, noi).clone())
        G["cases"].append(rec)
    out = OUT.parent / "flow_match_scheduler_vectors.pt"
    torch.save(G, out)
    print("wrote", out)


def gen_cache_names():
    """VAECache.generate_vae_cache_filename (caching/vae.py:678-703) and TextEmbeddingCache._normalize_key_value / create_hash
    (caching/text_embeds.py:126-154) lifted as methods and run on stub instances; base data backend's gzip container
    (data_backend/base.py:126-153) for one payload -> tests/golden/cache_io_vectors.pt"""
    import gzip
    import hashlib
    import os
    from enum import Enum
    from hashlib import sha256
    from io import BytesIO
    from types import SimpleNamespace
    gen, = extract(REF / "helpers" / "caching" / "vae.py", ["generate_vae_cache_filename"], class_name="VAECache", extra_ns={"os": os, "sha256": sha256})

    class Key(Enum):
        CAPTION = "caption"; FILENAME = "filename"; DATASET_AND_FILENAME = "dataset_and_filename"
    norm, mk = extract(REF / "helpers" / "caching" / "text_embeds.py", ["_normalize_key_value", "create_hash"], class_name="TextEmbeddingCache",
                       extra_ns={"os": os, "hashlib": hashlib, "TextEmbedCacheKey": Key, "canonicalize_data_uri": lambda x: x})
    G = {"vae": [], "text": []}
    for fp, cache_dir, inst, hashed in (("/data/imgs/cat.png", "/cache/vae", "/data/imgs", False), ("/data/imgs/sub/dir/dog.v2.jpeg", "/cache/vae", "/data/imgs", False),
                                        ("/data/imgs/sub/dog.jpg", "/cache/vae", "/data/imgs", True), ("/elsewhere/bird.webp", "/cache/vae", None, True),
                                        ("/cache/vae/already.pt", "/cache/vae", "/data/imgs", True)):
        stub = SimpleNamespace(image_data_backend=SimpleNamespace(), hash_filenames=hashed, cache_dir=cache_dir, instance_data_dir=inst)
        G["vae"].append((fp, cache_dir, inst, hashed, gen(stub, fp)))
    for key, prompt, model_type, path_based, key_type in (("a photo of a cat", "a photo of a cat", "flux", False, Key.CAPTION), ("", "", "sdxl", False, Key.CAPTION),
                                                          ("ünïcode ✓ caption", None, "sd3", False, Key.CAPTION), ("/data/imgs/cat.png", "a cat", "pixart_sigma", True, Key.FILENAME),
                                                          (None, None, "flux", False, Key.CAPTION)):
        stub = SimpleNamespace(model_type=model_type, key_type=key_type, _requires_path_based_keys=path_based)
        stub._normalize_key_value = lambda kv, _s=stub: norm(_s, kv)
        G["text"].append((key, prompt, model_type, path_based, key_type is Key.FILENAME, mk(stub, key, prompt=prompt)))
    payload = {"prompt_embeds": torch.arange(12, dtype=torch.float32).reshape(1, 3, 4).to(torch.bfloat16), "pooled_prompt_embeds": torch.ones(1, 4)}
    comp, = extract(REF / "helpers" / "data_backend" / "base.py", ["_compress_torch"], class_name="BaseDataBackend", extra_ns={"BytesIO": BytesIO, "gzip": gzip})
    G["gz_payload"], G["gz_bytes"] = payload, comp(SimpleNamespace(), payload)
    out = OUT.parent / "cache_io_vectors.pt"
    torch.save(G, out)
    print("wrote", out)


def _lora_key_cases():
    """state dicts in every dialect the converters see (shapes only matter through rank / in / out)"""
    g = torch.Generator().manual_seed(9)
    def AB(r, i, o):
        return torch.randn(r, i, generator=g), torch.randn(o, r, generator=g)
    peft_tr, peft_unet, old = {}, {}, {}
    for mod, r in (("transformer_blocks.0.attn.to_q", 4), ("transformer_blocks.0.attn.to_out.0", 4), ("single_transformer_blocks.3.attn.to_k", 8)):
        a, b = AB(r, 16, 16)
        peft_tr[f"transformer.{mod}.lora_A.weight"], peft_tr[f"transformer.{mod}.lora_B.weight"] = a, b
    for mod in ("down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q", "mid_block.attentions.0.transformer_blocks.0.attn2.processor.to_v"):
        a, b = AB(4, 16, 16)
        peft_unet[f"unet.{mod}.lora_A.weight"], peft_unet[f"unet.{mod}.lora_B.weight"] = a, b
        old[f"unet.{mod}.lora.down.weight"], old[f"unet.{mod}.lora.up.weight"] = a, b
    a, b = AB(4, 16, 16)
    peft_unet["text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_A.weight"], peft_unet["text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_B.weight"] = a, b
    a, b = AB(4, 16, 16)
    peft_unet["text_encoder_2.text_model.encoder.layers.1.mlp.fc1.lora_A.weight"], peft_unet["text_encoder_2.text_model.encoder.layers.1.mlp.fc1.lora_B.weight"] = a, b
    peft_unet["unet.some.buffer"] = torch.zeros(2)
    comfy = {}
    for mod, r, al in (("double_blocks.0.img_attn.qkv", 4, 8.0), ("single_blocks.1.linear1", 8, 8.0)):
        a, b = AB(r, 16, 16)
        comfy[f"diffusion_model.{mod}.lora_A.weight"], comfy[f"diffusion_model.{mod}.lora_B.weight"] = a, b
        comfy[f"diffusion_model.{mod}.alpha"] = torch.tensor(al)
    comfy["transformer.x.lora_A.weight"], comfy["bare.module.lora_B.weight"] = AB(4, 16, 16)
    return dict(peft_tr=peft_tr, peft_unet=peft_unet, old=old, comfy=comfy)


def gen_lora_keys():
    """helpers/training/lora_format.py loaded as a module where it lies (torch + stdlib only) and run over the dialect zoo above ->
    tests/golden/lora_keys_vectors.pt (keys, alpha values, shapes)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_lora_format", REF / "helpers" / "training" / "lora_format.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    Z = _lora_key_cases()
    meta = {"lora_alpha": 16, "alpha_pattern": {"single_transformer_blocks.3.attn.to_k": 2.0}}

    def shape_of(d):
        return {k: (tuple(v.shape), (float(v) if v.ndim == 0 else None)) for k, v in d.items()}
    G = {"detect": {n: (None if mod.detect_state_dict_format(d) is None else mod.detect_state_dict_format(d).value) for n, d in dict(Z, empty={}).items()},
         "normalize": {repr(v): mod.normalize_lora_format(v).value for v in (None, "", "ComfyUI ", "comfyui", "diffusers", "kohya", 3)},
         "ranks": {n: mod.collect_lora_ranks(d) for n, d in Z.items()},
         "ranks_stripped": mod.collect_lora_ranks(Z["peft_tr"], prefix_to_strip="transformer."),
         "alphas": {n: mod.collect_lora_alphas(d) for n, d in Z.items()},
         "synth": {n: mod.synthesize_missing_lora_alphas_from_ranks(d) for n, d in Z.items()},
         "synth_existing": mod.synthesize_missing_lora_alphas_from_ranks(Z["peft_tr"], existing_alphas={"x.alpha": 1.0}),
         "peft_kwargs": {n: mod.peft_lora_config_kwargs_from_state_dict(d) for n, d in Z.items()},
         "to_comfy": shape_of(mod.convert_diffusers_to_comfyui(Z["peft_tr"])),
         "to_comfy_keep_meta": shape_of(mod.convert_diffusers_to_comfyui(Z["peft_tr"], adapter_metadata=meta, preserve_component_prefixes={"transformer"})),
         "to_comfy_old": shape_of(mod.convert_diffusers_to_comfyui(Z["old"], adapter_metadata={"lora_alpha": torch.tensor(4.0)})),
         "to_kohya_sdxl": shape_of(mod.convert_diffusers_to_comfyui_sd_lora(Z["peft_unet"], adapter_metadata={"lora_alpha": 8},
                                                                             component_adapter_metadata={"text_encoder": {"lora_alpha": 2}}, sdxl=True)),
         "to_kohya_sd15": shape_of(mod.convert_diffusers_to_comfyui_sd_lora(Z["old"], sdxl=False)),
         "from_comfy": (lambda r: (shape_of(r[0]), r[1]))(mod.convert_comfyui_to_diffusers(Z["comfy"], target_prefix="transformer")),
         "from_comfy_noprefix": (lambda r: (shape_of(r[0]), r[1]))(mod.convert_comfyui_to_diffusers(Z["comfy"]))}
    conflict = {"m.lora_A.weight": torch.zeros(4, 8), "m.lora_B.weight": torch.zeros(8, 2)}
    try:
        mod.collect_lora_ranks(conflict)
        G["conflict"] = None
    except ValueError as e:
        G["conflict"] = str(e)
    out = OUT.parent / "lora_keys_vectors.pt"
    torch.save(G, out)
    print("wrote", out)


def gen_lr_schedules():
    """Cosine / CosineAnnealingHardRestarts / Sine / get_polynomial_decay_schedule_with_warmup (helpers/training/custom_schedule.py:102-440) lifted by
    AST and stepped on a real torch optimizer -> tests/golden/lr_schedule_vectors.pt (the learning rate after every step)"""
    import logging
    from torch.optim.lr_scheduler import LambdaLR, LRScheduler
    path = REF / "helpers" / "training" / "custom_schedule.py"
    tree = ast.parse(path.read_text())
    want = {"_enable_get_lr_call", "Cosine", "CosineAnnealingHardRestarts", "Sine", "get_polynomial_decay_schedule_with_warmup"}
    picked = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in want]
    assert {n.name for n in picked} == want
    ns = {"torch": torch, "math": math, "LambdaLR": LambdaLR, "LRScheduler": LRScheduler, "logger": logging.getLogger("ref"), "__builtins__": __builtins__}
    modl = ast.Module(body=picked, type_ignores=[])
    ast.fix_missing_locations(modl)
    exec(compile(modl, str(path), "exec"), ns)

    def run(make, n):
        p_ = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p_], lr=1e-4)
        sch = make(opt)
        out = [opt.param_groups[0]["lr"]]
        for _ in range(n):
            opt.step()
            sch.step()
            out.append(opt.param_groups[0]["lr"])
        return out
    G = {}
    for T0, eta in ((10, 0.0), (7, 1e-6), (100, 1e-7)):
        G[("sine", T0, eta)] = run(lambda o: ns["Sine"](optimizer=o, T_0=T0, T_mult=1, eta_min=eta, last_step=-1), 45)
        G[("cosine", T0, eta)] = run(lambda o: ns["Cosine"](optimizer=o, T_0=T0, T_mult=1, eta_min=eta, last_step=-1), 45)
        G[("cosine_with_restarts", T0, eta)] = run(lambda o: ns["CosineAnnealingHardRestarts"](optimizer=o, T_0=T0, T_mult=1, eta_min=eta, last_step=-1), 45)
    for warm, total, end, power in ((5, 30, 1e-7, 1.0), (0, 20, 1e-6, 2.0), (10, 25, 1e-8, 0.5)):
        G[("polynomial", warm, total, end, power)] = run(lambda o: ns["get_polynomial_decay_schedule_with_warmup"](optimizer=o, num_warmup_steps=warm, num_training_steps=total,
                                                                                                               lr_end=end, power=power, last_epoch=-1), 40)
    out = OUT.parent / "lr_schedule_vectors.pt"
    torch.save(G, out)
    print("wrote", out)


def main():
    gen_lr_schedules()
    gen_lora_keys()
    gen_cache_names()
    gen_flow_match_scheduler()
    gen_collate()
    gen_cubic_schedule()
    gen_adamw_bf16()
    gen_loss()
    gen_fp8()
    gen_ddpm_sampling()
    torch.manual_seed(1234)
    G = {}
    cite = {}

    # ---- Flux pack / unpack / ids (flux/__init__.py:25-63) ----
    p = REF / "helpers/models/flux/__init__.py"
    pack_latents, unpack_latents, prepare_latent_image_ids = extract(p, ["pack_latents", "unpack_latents", "prepare_latent_image_ids"])
    lat = torch.randn(2, 16, 12, 20).to(torch.bfloat16)
    packed = pack_latents(lat, 2, 16, 12, 20)
    G["pack.in"] = lat; G["pack.out"] = packed
    G["unpack.out"] = unpack_latents(packed, 12 * 8, 20 * 8, 16)
    G["ids.12x20"] = prepare_latent_image_ids(2, 12, 20, "cpu", torch.float32)
    cite["pack"] = "simpletuner/helpers/models/flux/__init__.py:25-63"

    # ---- RoPE application (flux/transformer.py:73-98) ----
    p = REF / "helpers/models/flux/transformer.py"
    (apply_rope,) = extract(p, ["_apply_rotary_emb_anyshape"])
    S, d = 24, 128
    ang = torch.rand(S, d // 2, dtype=torch.float64) * 6
    cos = ang.cos().repeat_interleave(2, dim=1).float(); sin = ang.sin().repeat_interleave(2, dim=1).float()
    x = torch.randn(2, 3, S, d).to(torch.bfloat16)
    G["rope.x"] = x; G["rope.cos"] = cos; G["rope.sin"] = sin
    G["rope.out_bf16"] = apply_rope(x, (cos, sin))
    G["rope.out_fp32"] = apply_rope(x.float(), (cos, sin))
    cite["rope"] = "simpletuner/helpers/models/flux/transformer.py:73-98"

    # ---- flow schedule shift (custom_schedule.py:443-478) ----
    p = REF / "helpers/training/custom_schedule.py"
    (shift_fn,) = extract(p, ["apply_flow_schedule_shift"], extra_ns={"calculate_shift_flux": None})
    sig = torch.tensor([0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99])
    for sh in (1.0, 3.0, 0.5):
        args = types.SimpleNamespace(flow_schedule_shift=sh, flow_schedule_auto_shift=False)
        G[f"shift.{sh}"] = shift_fn(args, None, sig.clone(), torch.zeros(1, 16, 8, 8))
    G["shift.in"] = sig
    args = types.SimpleNamespace(flow_schedule_shift=None, flow_schedule_auto_shift=False)
    G["shift.none"] = shift_fn(args, None, sig.clone(), torch.zeros(1, 16, 8, 8))
    cite["shift"] = "simpletuner/helpers/training/custom_schedule.py:443-478"

    # ---- EMA decay schedule (ema.py:322-349) ----
    p = REF / "helpers/training/ema.py"
    (get_decay,) = extract(p, ["get_decay"], class_name="EMAModel")
    rows = []
    for (decay, min_decay, uas, warm, use_w, inv_g, power) in [(0.9999, 0.0, 0, 0, False, 1.0, 2 / 3), (0.999, 0.0, 0, 0, False, 1.0, 2 / 3),
                                                              (0.9, 0.0, 0, 3, False, 1.0, 2 / 3), (0.9999, 0.5, 10, 0, True, 1.0, 0.75)]:
        self_ = types.SimpleNamespace(decay=decay, min_decay=min_decay, update_after_step=uas, warmup_steps=warm, use_ema_warmup=use_w,
                                      inv_gamma=inv_g, power=power, optimization_step=0)
        for step in (0, 1, 2, 3, 4, 10, 11, 12, 100, 1000, 100000):
            rows.append([decay, min_decay, uas, warm, float(use_w), inv_g, power, step, get_decay(self_, step)])
    G["ema.decay_table"] = torch.tensor(rows, dtype=torch.float64)
    # update formula ema.py:423 (foreach) / :430 (loop): s -= (1-d)(s-p); known answers from tests/test_ema.py:39-105
    s0 = torch.randn(257); pp = torch.randn(257)
    s_foreach = [s0.clone()]
    torch._foreach_sub_(s_foreach, torch._foreach_sub(s_foreach, [pp]), alpha=1 - 0.999)
    G["ema.s0"] = s0; G["ema.p"] = pp; G["ema.s1_decay0.999"] = s_foreach[0]
    cite["ema"] = "simpletuner/helpers/training/ema.py:322-349, 423"

    # ---- sample-weighted loss gather (context_parallel_sync.py:327-348) ----
    p = REF / "helpers/data_backend/runtime/context_parallel_sync.py"
    import numbers
    fns = extract(p, ["_normalize_parallel_size", "gather_sample_weighted_scalar"], extra_ns={"numbers": numbers})

    class FakeAcc:
        num_processes = 2

        def gather(self, t):
            # rank 0: loss 2.0 with 1 sample ; rank 1: loss 4.0 with 3 samples  -> weighted mean 3.5 (tests/test_distributed_batch_layout.py:224-235)
            return torch.stack([torch.tensor([2.0 * 1, 1.0]), torch.tensor([4.0 * 3, 3.0])]).reshape(-1)

    G["gather.weighted"] = fns[1](torch.tensor(2.0), 1, FakeAcc())
    cite["gather"] = "simpletuner/helpers/data_backend/runtime/context_parallel_sync.py:327-348"

    # ---- flow noising / target known answers (tests/test_flux_model.py:122-124, tests/test_mixflow.py:45-92) ----
    # the reference formulas are one-liners inside methods with heavy `self`; pinned by their own tests' expectations:
    x0 = torch.randn(2, 16, 4, 4); n0 = torch.randn(2, 16, 4, 4)
    G["flow.x"] = x0; G["flow.n"] = n0
    G["flow.noisy_sigma0.25"] = 0.75 * x0 + 0.25 * n0       # expected_noisy, tests/test_flux_model.py:122
    G["flow.target"] = n0 - x0                              # tests/test_flux_model.py:124
    cite["flow"] = "tests/test_flux_model.py:122-124 (expected values of common.py:4990, 4610-4611)"

    G["_cite"] = cite
    OUT.parent.mkdir(parents=True, exist_ok=True)
    torch.save(G, OUT)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes): {len(G) - 1} tensors")


if __name__ == "__main__":
    if not REF.exists():
        sys.exit("reference tree not present (this script only runs in the build container)")
    main()
