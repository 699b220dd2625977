"""The oracle restatements of the three transformer families, PINNED to outputs of the reference's own model files executed in the build
container (tools/gen_ref_models.py through tools/ref_shim.py -> tests/golden/ref_{flux,sd3,pixart}_model.pt).

What is reference code in those fixtures: FluxTransformer2DModel.forward / FluxTransformerBlock / FluxSingleTransformerBlock / FluxAttnProcessor2_0 /
_apply_rotary_emb_anyshape / expand_flux_attention_mask (flux/transformer.py:73-1513), SD3Transformer2DModel.forward and
_sd3_apply_joint_transformer_block incl. SD3.5 dual attention (sd3/transformer.py:126-241, 560-911), PixArtTransformer2DModel.forward and the
tokenwise block (pixart/transformer.py:95-145, 499-788), the ControlNet-Transformer wrapper (pixart/controlnet.py:13-326), TREADRouter
(training/tread.py) and the checkpoint planners (training/gradient_checkpointing_interval.py).  Leaf modules (Linear / LayerNorm / SiLU compositions
of diffusers) are shims, partly lifted from in-tree vendored copies (tools/ref_shim.py header).

Tolerance: fp32 vs fp32, rel-L2 <= 1e-5 on outputs and on every gradient (different summation orders only)."""
import os

import pytest
import torch

from oracle import flux as OF
from oracle import pixart as OP
from oracle import sd3 as OS
from tests.ref_fixture_utils import rel_l2, seeded_lora, seeded_state, state_checksum

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5


def _load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def _state(shapes, seed, checksum, bf16=False):
    st = seeded_state(shapes, seed)
    if bf16:
        st = {k: v.to(torch.bfloat16).float() for k, v in st.items()}
    cs = state_checksum(st)
    assert abs(cs - checksum) <= 1e-6 * max(1.0, abs(checksum)), f"the seeded weights differ from the ones the reference ran with: {cs} vs {checksum}"
    return st


def _backward(out, w, leaves):
    for t in leaves:
        t.grad = None
    (out * w).sum().backward()


def _check_grads(P, ref_grads, tag):
    worst = (0.0, "")
    for n, g in ref_grads.items():
        assert P[n].grad is not None, f"{tag}: no gradient for {n}"
        if float(g.double().norm()) <= 1e-5 * g.numel() ** 0.5:
            # analytically-zero gradients (a key bias shifts every score of a query alike: softmax is invariant): both sides hold rounding noise only
            assert float(P[n].grad.double().norm()) <= 1e-4 * g.numel() ** 0.5, f"{tag}: {n} should be ~0"
            continue
        worst = max(worst, (rel_l2(P[n].grad, g), n))
    assert worst[0] <= TOL, f"{tag}: gradient {worst[1]} rel-L2 {worst[0]:.3e}"
    return worst


# ------------------------------------------------------------------------------------------------------------------------
def _flux_cfg(c):
    return OF.FluxConfig(in_channels=c["in_channels"], num_layers=c["num_layers"], num_single_layers=c["num_single_layers"],
                         attention_head_dim=c["attention_head_dim"], num_attention_heads=c["num_attention_heads"],
                         joint_attention_dim=c["joint_attention_dim"], pooled_projection_dim=c["pooled_projection_dim"],
                         guidance_embeds=c["guidance_embeds"], axes_dims_rope=tuple(c["axes_dims_rope"]))


def _flux_run(P, cfg, inputs, **kw):
    leaves = {k: inputs[k].clone().requires_grad_(True) for k in ("hidden_states", "encoder_hidden_states", "pooled_projections")}
    out = OF.flux_forward(P, cfg, leaves["hidden_states"], leaves["encoder_hidden_states"], leaves["pooled_projections"], inputs["timestep"],
                          inputs["img_ids"], inputs["txt_ids"], inputs["guidance"], **kw)
    return out, leaves


@pytest.mark.parametrize("case", ["plain", "masked", "tread_double", "tread_single"])
def test_flux_oracle_reproduces_reference_model(case):
    G = _load("ref_flux_model.pt")["tiny"]
    cfg = _flux_cfg(G["config"])
    P = {k: v.requires_grad_(True) for k, v in _state(OF.param_shapes(cfg), G["seed"], G["state_checksum"]).items()}
    R = G["cases"][case]
    kw = {}
    if case == "masked":
        # the reference hands SDPA `(mask > 0).bool().to(dtype)`: a FLOAT mask, i.e. +1.0 on kept keys and +0.0 on masked ones (flux/transformer.py:170-173)
        am = R["attention_mask"]
        kb = torch.ones(am.shape[0], am.shape[1] + G["inputs"]["hidden_states"].shape[1])
        kb[:, : am.shape[1]] = (am > 0).float()
        kw["key_bias"] = kb
    if case.startswith("tread"):
        kw["tread"] = {"routes": R["routes"], "mask_infos": R["mask_infos"]}
    out, leaves = _flux_run(P, cfg, G["inputs"], **kw)
    r = rel_l2(out, R["out"])
    assert r <= TOL, f"flux {case}: output rel-L2 {r:.3e}"
    _backward(out, R["w"], list(P.values()) + list(leaves.values()))
    worst = _check_grads(P, R["grads"], f"flux {case}")
    for k, g in R["input_grads"].items():
        if k in leaves:
            assert rel_l2(leaves[k].grad, g) <= TOL, (case, k)
    print(f"[pinned] flux {case}: out rel-L2 {r:.2e}, worst of {len(R['grads'])} parameter gradients {worst[0]:.2e} ({worst[1]})")


def test_flux_oracle_reproduces_reference_model_with_tokenwise_timesteps():
    """TOKENWISE timesteps [B, S_img] (CREPA self-flow; reference tests/test_flux_model.py:213-241): the reference's FluxTransformer2DModel executed with one timestep
    per image token (tools/gen_ref_tokenwise.py) — per-token AdaLN rows on the image stream of the double blocks, the token mean on the text stream, [mean x S_txt ||
    per token] along the single blocks' joint sequence, per-token rows in norm_out; output and EVERY parameter / input gradient of oracle.flux to <= 1e-5"""
    G = _load("ref_tokenwise.pt")["flux"]
    cfg = _flux_cfg(G["config"])
    P = {k: v.requires_grad_(True) for k, v in _state(OF.param_shapes(cfg), G["seed"], G["state_checksum"]).items()}
    R = G["case"]
    assert R["inputs"]["timestep"].ndim == 2
    out, leaves = _flux_run(P, cfg, R["inputs"])
    r = rel_l2(out, R["out"])
    assert r <= TOL, f"flux tokenwise: output rel-L2 {r:.3e}"
    _backward(out, R["w"], list(P.values()) + list(leaves.values()))
    worst = _check_grads(P, R["grads"], "flux tokenwise")
    for k, g in R["input_grads"].items():
        if k in leaves:
            assert rel_l2(leaves[k].grad, g) <= TOL, k
    print(f"[pinned] flux tokenwise: out rel-L2 {r:.2e}, worst of {len(R['grads'])} parameter gradients {worst[0]:.2e} ({worst[1]})")


@pytest.mark.parametrize("which", ["all", "context", "all+ffs", "context+ffs", "all+ffs+embedder", "ai-toolkit", "nano", "tiny"])
def test_flux_oracle_lora_target_sets_reproduce_the_reference_model_with_merged_adapters(which):
    """`flux_lora_target` sets (flux/model.py:1235-1380) in oracle.flux — adapters as separate factors on `lora_targets(cfg, which)` — against the reference's
    FluxTransformer2DModel executed with the MERGED weights W' = W + (alpha / r) B A on the modules peft's suffix rule selects from the reference's OWN lists
    (tools/gen_ref_flux_lora_sets.py reads them with `ast`): the same module set (note "proj_out" in all+ffs also names the model's output projection), the same
    output and input gradients, and the adapter gradients dL/dW' implies"""
    G = _load("ref_flux_lora_sets.pt")[which]
    cfg = _flux_cfg(G["config"])
    shapes = OF.param_shapes(cfg)
    P = _state(shapes, G["seed"], G["state_checksum"])
    assert sorted(OF.lora_targets(cfg, which)) == sorted(G["lora_targets"])
    lora = {k: (a.requires_grad_(True), b.requires_grad_(True)) for k, (a, b) in seeded_lora(G["lora_targets"], shapes, G["lora_rank"], G["lora_seed"]).items()}
    out, leaves = _flux_run(P, cfg, G["inputs"], lora=lora, lora_scale=G["lora_alpha"] / G["lora_rank"])
    r = rel_l2(out, G["out"])
    assert r <= TOL, f"flux {which}: output rel-L2 {r:.3e}"
    (out * G["w"]).sum().backward()
    for k, g in G["input_grads"].items():
        if k in leaves:
            assert rel_l2(leaves[k].grad, g) <= TOL, (which, k)
    worst = (0.0, "")
    for k, (dA, dB) in G["lora_grads"].items():
        ra, rb = rel_l2(lora[k][0].grad, dA), rel_l2(lora[k][1].grad, dB)
        worst = max(worst, (ra, k + ".A"), (rb, k + ".B"))
        assert ra <= 5 * TOL and rb <= 5 * TOL, (which, k, ra, rb)
    print(f"[pinned] flux lora target set {which}: {len(G['lora_targets'])} wrapped modules, out rel-L2 {r:.2e}, worst adapter gradient {worst[0]:.2e} ({worst[1]})")


def test_flux_oracle_reproduces_reference_model_at_kernel_head_width():
    """the "hip" tier (2 heads x 128, LoRA r4): weights rebuilt from the seed, LoRA applied as an adapter in the oracle; the reference ran the MERGED weight"""
    G = _load("ref_flux_model.pt")["hip"]
    cfg = _flux_cfg(G["config"])
    shapes = OF.param_shapes(cfg)
    P = _state(shapes, G["seed"], G["state_checksum"], bf16=True)
    lora = {k: (a.requires_grad_(True), b.requires_grad_(True)) for k, (a, b) in seeded_lora(G["lora_targets"], shapes, G["lora_rank"], G["lora_seed"]).items()}
    out, leaves = _flux_run(P, cfg, G["inputs"], lora=lora, lora_scale=G["lora_alpha"] / G["lora_rank"])
    assert rel_l2(out, G["out"]) <= TOL
    (out * G["w"]).sum().backward()
    for k, (dA, dB) in G["lora_grads"].items():
        assert rel_l2(lora[k][0].grad, dA) <= 5 * TOL and rel_l2(lora[k][1].grad, dB) <= 5 * TOL, k
    for k, g in G["input_grads"].items():
        if k in leaves:
            assert rel_l2(leaves[k].grad, g) <= TOL, k


def test_checkpoint_plans_match_the_blocks_the_reference_wraps():
    """flux/transformer.py:1142-1209, 1243-1290 and sd3/transformer.py:716-833 decide per mode which blocks run under the checkpoint function; the
    fixture recorded them by intercepting that function.  simpletuner_amd.training.checkpoint_plan must produce the same segments."""
    from simpletuner_amd.training import checkpoint_plan as CP

    F_ = _load("ref_flux_model.pt")["tiny"]
    nd, ns = F_["config"]["num_layers"], F_["config"]["num_single_layers"]
    for tag, plan in F_["checkpoint_plans"].items():
        want_d = [[int(n[1:]) for n in seg] for seg in plan["wrapped"] if seg[0][0] == "d"]
        want_s = [[int(n[1:]) for n in seg] for seg in plan["wrapped"] if seg[0][0] == "s"]
        got_d = CP.flux_segments(nd, plan["interval"], plan["stride"])
        got_s = CP.flux_segments(ns, plan["interval"], plan["stride"])
        assert got_d == want_d and got_s == want_s, (tag, got_d, want_d, got_s, want_s)
    S_ = _load("ref_sd3_model.pt")["tiny"]["sd3"]
    n = S_["config"]["num_layers"]
    for tag, plan in S_["checkpoint_plans"].items():
        want = [[int(b[1:]) for b in seg] for seg in plan["wrapped"]]
        got = CP.sd3_segments(n, plan["interval"], plan["stride"])
        assert got == want, (tag, got, want)


# ------------------------------------------------------------------------------------------------------------------------
def _sd3_cfg(c):
    return OS.SD3Config(sample_size=c["sample_size"], patch_size=c["patch_size"], in_channels=c["in_channels"], num_layers=c["num_layers"],
                        attention_head_dim=c["attention_head_dim"], num_attention_heads=c["num_attention_heads"], joint_attention_dim=c["joint_attention_dim"],
                        pooled_projection_dim=c["pooled_projection_dim"], out_channels=c["out_channels"], pos_embed_max_size=c["pos_embed_max_size"],
                        qk_norm=c.get("qk_norm"), dual_attention_layers=tuple(c.get("dual_attention_layers", ())))


def _sd3_params(cfg, seed, checksum, bf16=False):
    P = _state(OS.param_shapes(cfg), seed, checksum, bf16)
    P["pos_embed.pos_embed"] = OS.sincos_2d(cfg.inner_dim, cfg.pos_embed_max_size, cfg.sample_size // cfg.patch_size)[None]
    return P


@pytest.mark.parametrize("variant,case", [("sd3", "wide"), ("sd3", "square"), ("sd3", "tall"), ("sd3", "tread"), ("sd35", "wide")])
def test_sd3_oracle_reproduces_reference_model(variant, case):
    V = _load("ref_sd3_model.pt")["tiny"][variant]
    cfg = _sd3_cfg(V["config"])
    P = _sd3_params(cfg, V["seed"], V["state_checksum"])
    assert rel_l2(P["pos_embed.pos_embed"], V["pos_embed_table"]) <= 1e-6          # PatchEmbed's sincos table (diffusers formula) as the shim built it
    P = {k: (v.requires_grad_(True) if k != "pos_embed.pos_embed" else v) for k, v in P.items()}
    R = V["cases"][case]
    I = R["inputs"]
    leaves = {k: I[k].clone().requires_grad_(True) for k in ("hidden_states", "encoder_hidden_states", "pooled_projections")}
    kw = {"tread": {"routes": R["routes"], "mask_infos": R["mask_infos"]}} if case == "tread" else {}
    out = OS.sd3_forward(P, cfg, leaves["hidden_states"], leaves["encoder_hidden_states"], leaves["pooled_projections"], I["timestep"], **kw)
    r = rel_l2(out, R["out"])
    assert r <= TOL, f"sd3 {variant}/{case}: output rel-L2 {r:.3e}"
    (out * R["w"]).sum().backward()
    worst = _check_grads(P, R["grads"], f"sd3 {variant}/{case}")
    for k, g in R["input_grads"].items():
        if k in leaves:
            assert rel_l2(leaves[k].grad, g) <= TOL, (variant, case, k)
    print(f"[pinned] sd3 {variant}/{case}: out rel-L2 {r:.2e}, worst of {len(R['grads'])} parameter gradients {worst[0]:.2e} ({worst[1]})")


def test_sd3_oracle_reproduces_reference_model_with_tokenwise_timesteps():
    """TOKENWISE timesteps [B, S_img] (CREPA self-flow; reference tests/test_sd3_model.py:179-204): the reference's SD3Transformer2DModel executed with one
    timestep per image token (tools/gen_ref_tokenwise.py) — per-token AdaLN rows on the image stream and in norm_out, their mean on the context stream; output
    and EVERY parameter / input gradient of oracle.sd3 to <= 1e-5"""
    V = _load("ref_tokenwise.pt")["sd3"]
    cfg = _sd3_cfg(V["config"])
    P = _sd3_params(cfg, V["seed"], V["state_checksum"])
    assert rel_l2(P["pos_embed.pos_embed"], V["pos_embed_table"]) <= 1e-6
    P = {k: (v.requires_grad_(True) if k != "pos_embed.pos_embed" else v) for k, v in P.items()}
    R = V["case"]
    I = R["inputs"]
    assert I["timestep"].ndim == 2
    leaves = {k: I[k].clone().requires_grad_(True) for k in ("hidden_states", "encoder_hidden_states", "pooled_projections")}
    out = OS.sd3_forward(P, cfg, leaves["hidden_states"], leaves["encoder_hidden_states"], leaves["pooled_projections"], I["timestep"])
    r = rel_l2(out, R["out"])
    assert r <= TOL, f"sd3 tokenwise: output rel-L2 {r:.3e}"
    (out * R["w"]).sum().backward()
    worst = _check_grads(P, R["grads"], "sd3 tokenwise")
    for k, g in R["input_grads"].items():
        if k in leaves:
            assert rel_l2(leaves[k].grad, g) <= TOL, k
    print(f"[pinned] sd3 tokenwise: out rel-L2 {r:.2e}, worst of {len(R['grads'])} parameter gradients {worst[0]:.2e} ({worst[1]})")


@pytest.mark.parametrize("variant", ["sd3", "sd35"])
def test_sd3_oracle_reproduces_reference_model_at_kernel_head_width(variant):
    H = _load("ref_sd3_model.pt")["hip"][variant]
    cfg = _sd3_cfg(H["config"])
    P = _sd3_params(cfg, H["seed"], H["state_checksum"], bf16=True)
    P = {k: (v.requires_grad_(True) if k != "pos_embed.pos_embed" else v) for k, v in P.items()}
    I = H["inputs"]
    out = OS.sd3_forward(P, cfg, I["hidden_states"], I["encoder_hidden_states"], I["pooled_projections"], I["timestep"])
    assert rel_l2(out, H["out"]) <= TOL
    (out * H["w"]).sum().backward()
    _check_grads(P, H["full_ft_grads"], f"sd3 hip {variant}")
    if "lora" in H:
        L = H["lora"]
        shapes = OS.param_shapes(cfg)
        lora = {k: (a.requires_grad_(True), b.requires_grad_(True)) for k, (a, b) in seeded_lora(L["lora_targets"], shapes, L["lora_rank"], L["lora_seed"]).items()}
        Pd = {k: v.detach() for k, v in P.items()}
        out = OS.sd3_forward(Pd, cfg, I["hidden_states"], I["encoder_hidden_states"], I["pooled_projections"], I["timestep"], lora=lora,
                             lora_scale=L["lora_alpha"] / L["lora_rank"])
        assert rel_l2(out, L["out"]) <= TOL
        (out * H["w"]).sum().backward()
        for k, (dA, dB) in L["lora_grads"].items():
            assert rel_l2(lora[k][0].grad, dA) <= 5 * TOL and rel_l2(lora[k][1].grad, dB) <= 5 * TOL, k


# ------------------------------------------------------------------------------------------------------------------------
def _pix_cfg(c):
    return OP.PixArtConfig(num_attention_heads=c["num_attention_heads"], attention_head_dim=c["attention_head_dim"], in_channels=c["in_channels"],
                           out_channels=c["out_channels"], num_layers=c["num_layers"], cross_attention_dim=c["cross_attention_dim"],
                           sample_size=c["sample_size"], patch_size=c["patch_size"], caption_channels=c["caption_channels"],
                           use_additional_conditions=c["use_additional_conditions"])


def _adapter_shapes(cfg, n_ctrl):
    D = cfg.D
    one = {k[len("transformer_blocks.0."):]: v for k, v in OP.param_shapes(cfg).items() if k.startswith("transformer_blocks.0.")}
    sh = {}
    for i in range(n_ctrl):
        p = f"controlnet_blocks.{i}."
        if i == 0:
            sh[p + "before_proj.weight"], sh[p + "before_proj.bias"] = (D, D), (D,)
        sh[p + "after_proj.weight"], sh[p + "after_proj.bias"] = (D, D), (D,)
        for k, v in one.items():
            sh[p + "transformer_block." + k] = v
    return sh


def test_pixart_oracle_reproduces_reference_model_with_a_tread_route():
    """the reference's PixArtTransformer2DModel EXECUTED in train mode with a TREAD router (pixart/transformer.py:487-489, 677-741; tools/gen_ref_tokenwise.py::gen_pixart_tread,
    the router's permutation recorded): oracle.pixart with the same route replayed — output and every parameter / input gradient to <= 1e-5"""
    G = _load("ref_tokenwise.pt")["pixart_tread"]
    cfg = _pix_cfg(G["config"])
    P = {k: v.requires_grad_(True) for k, v in _state(OP.param_shapes(cfg), G["seed"], G["state_checksum"], False).items()}
    R = G["case"]
    I = R["inputs"]
    lat = I["hidden_states"].clone().requires_grad_(True)
    out = OP.pixart_forward(P, cfg, lat, I["encoder_hidden_states"], I["encoder_attention_mask"], I["timestep"], I["resolution"], I["aspect_ratio"],
                            tread={"routes": R["routes"], "mask_infos": R["mask_infos"]})
    r = rel_l2(out, R["out"])
    assert r <= TOL, f"pixart tread: rel-L2 {r:.3e}"
    (out * R["w"]).sum().backward()
    worst = _check_grads(P, R["grads"], "pixart tread")
    assert rel_l2(lat.grad, R["input_grads"]["hidden_states"]) <= TOL
    print(f"[pinned] pixart TREAD route: out rel-L2 {r:.2e}, worst of {len(R['grads'])} parameter gradients {worst[0]:.2e} ({worst[1]})")


def test_pixart_oracle_lora_reproduces_the_reference_model_with_merged_adapters():
    """PixArt LoRA (pixart/model.py:59 targets) in oracle.pixart — adapters as separate factors — against the reference trunk executed with the MERGED weights
    W' = W + (alpha / r) B A: same output, same input gradient, and the adapter gradients dL/dW' implies"""
    G = _load("ref_tokenwise.pt")["pixart_lora"]
    cfg = _pix_cfg(G["config"])
    shapes = OP.param_shapes(cfg)
    P = _state(shapes, G["seed"], G["state_checksum"], False)
    lora = {k: (a.requires_grad_(True), b.requires_grad_(True)) for k, (a, b) in seeded_lora(G["lora_targets"], shapes, G["lora_rank"], G["lora_seed"]).items()}
    assert sorted(lora) == sorted(OP.lora_targets(cfg))
    I = G["inputs"]
    lat = I["hidden_states"].clone().requires_grad_(True)
    out = OP.pixart_forward(P, cfg, lat, I["encoder_hidden_states"], I["encoder_attention_mask"], I["timestep"], I["resolution"], I["aspect_ratio"], lora=lora,
                            lora_scale=G["lora_alpha"] / G["lora_rank"])
    assert rel_l2(out, G["out"]) <= TOL
    (out * G["w"]).sum().backward()
    assert rel_l2(lat.grad, G["input_grads"]["hidden_states"]) <= TOL
    for k, (dA, dB) in G["lora_grads"].items():
        assert rel_l2(lora[k][0].grad, dA) <= 5 * TOL and rel_l2(lora[k][1].grad, dB) <= 5 * TOL, k


def test_pixart_oracle_reproduces_reference_model_with_tokenwise_timesteps():
    """TOKENWISE timesteps [B, S] (CREPA self-flow; reference tests/test_pixart_model.py:91-115): the reference's PixArtTransformer2DModel executed with one timestep
    per token (tools/gen_ref_tokenwise.py) — per-token AdaLN-single rows in every block and in the head, the size conditions shared by a sample's tokens;
    output and every parameter / input gradient of oracle.pixart to <= 1e-5"""
    G = _load("ref_tokenwise.pt")["pixart"]
    cfg = _pix_cfg(G["config"])
    P = {k: v.requires_grad_(True) for k, v in _state(OP.param_shapes(cfg), G["seed"], G["state_checksum"], False).items()}
    R = G["case"]
    I = R["inputs"]
    assert I["timestep"].ndim == 2
    lat = I["hidden_states"].clone().requires_grad_(True)
    out = OP.pixart_forward(P, cfg, lat, I["encoder_hidden_states"], I["encoder_attention_mask"], I["timestep"], I["resolution"], I["aspect_ratio"])
    r = rel_l2(out, R["out"])
    assert r <= TOL, f"pixart tokenwise: rel-L2 {r:.3e}"
    (out * R["w"]).sum().backward()
    worst = _check_grads(P, R["grads"], "pixart tokenwise")
    assert rel_l2(lat.grad, R["input_grads"]["hidden_states"]) <= TOL
    print(f"[pinned] pixart tokenwise: out rel-L2 {r:.2e}, worst of {len(R['grads'])} parameter gradients {worst[0]:.2e} ({worst[1]})")


@pytest.mark.parametrize("tier", ["tiny", "hip"])
def test_pixart_oracle_reproduces_reference_trunk_and_controlnet(tier):
    G = _load("ref_pixart_model.pt")[tier]
    cfg = _pix_cfg(G["config"])
    bf16 = tier == "hip"
    P = {k: v.requires_grad_(True) for k, v in _state(OP.param_shapes(cfg), G["seed"], G["state_checksum"], bf16).items()}
    C = {k: v.requires_grad_(True) for k, v in _state(_adapter_shapes(cfg, G["n_ctrl"]), G["adapter_seed"], G["adapter_checksum"], bf16).items()}
    I = G["inputs"]
    lat = I["hidden_states"].clone().requires_grad_(True)
    out = OP.pixart_forward(P, cfg, lat, I["encoder_hidden_states"], I["encoder_attention_mask"], I["timestep"], I["resolution"], I["aspect_ratio"])
    ref_out, ref_w = (G["cases"]["trunk"]["out"], G["cases"]["trunk"]["w"]) if tier == "tiny" else (G["trunk_out"], G["trunk_w"])
    r = rel_l2(out, ref_out)
    assert r <= TOL, f"pixart trunk {tier}: rel-L2 {r:.3e}"
    (out * ref_w).sum().backward()
    if tier == "tiny":
        _check_grads(P, G["cases"]["trunk"]["grads"], "pixart trunk")
        assert rel_l2(lat.grad, G["cases"]["trunk"]["input_grads"]["hidden_states"]) <= TOL
    else:
        assert rel_l2(lat.grad, G["trunk_input_grads"]["hidden_states"]) <= TOL
    for t in list(P.values()):
        t.grad = None
    cond = I["controlnet_cond"].clone().requires_grad_(True)
    out = OP.controlnet_forward(P, C, cfg, G["n_ctrl"], I["hidden_states"], cond, I["encoder_hidden_states"], I["encoder_attention_mask"], I["timestep"],
                                I["resolution"], I["aspect_ratio"])
    R = G["cases"]["controlnet"] if tier == "tiny" else G
    r = rel_l2(out, R["out"])
    assert r <= TOL, f"pixart controlnet {tier}: rel-L2 {r:.3e}"
    (out * R["w"]).sum().backward()
    worst = _check_grads(C, R["grads"] if tier == "tiny" else R["adapter_grads"], f"pixart controlnet {tier}")
    assert rel_l2(cond.grad, R["input_grads"]["controlnet_cond"]) <= TOL
    print(f"[pinned] pixart {tier}: controlnet out rel-L2 {r:.2e}, worst adapter gradient {worst[0]:.2e} ({worst[1]})")


def test_pixart_oracle_block_equals_the_reference_tokenwise_block():
    """pixart/transformer.py:95-145 executed on a [B, S, 6D] broadcast of the batch-wise modulation vs oracle.pixart.block"""
    G = _load("ref_pixart_model.pt")["tiny"]
    cfg = _pix_cfg(G["config"])
    P = _state(OP.param_shapes(cfg), G["seed"], G["state_checksum"])
    bc = G["block_case"]
    out = OP.block(P, f"transformer_blocks.{bc['block_index']}.", cfg, bc["h"], bc["ctx"], bc["bias"][:, 0, :], bc["t6"])
    assert rel_l2(out, bc["out_tokenwise_reference_code"]) <= TOL


# ------------------------------------------------------------------------------------------------------------------------
# VAE: the KL autoencoder the reference vendors together with a diffusers-key converter (tools/gen_ref_models.py gen_vae)
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["sdxl_layout", "flux_layout"])
def test_vae_oracle_reproduces_the_reference_autoencoder(case):
    """oracle/vae.py (AutoencoderKL.encode -> moments, .decode -> pixels, under diffusers' key names) against the reference's own vendored KL autoencoder
    EXECUTED on the same seeded checkpoint (its `convert_diffusers_state_dict` accepted exactly the oracle's key names, its `load_state_dict` their shapes):
    moments, pixels and the input gradients of both halves, fp32, <= 1e-5."""
    from oracle import vae as OV
    G = torch.load(os.path.join(GOLD, "ref_vae_model.pt"))
    c = G["cases"][case]
    cfg = OV.VAEConfig(latent_channels=c["latent_channels"], block_out_channels=tuple(c["block_out_channels"]), use_quant_conv=c["use_quant_conv"])
    shapes = {k: tuple(v.shape) for k, v in OV.init_params(cfg, shapes_only=True).items()}
    st = seeded_state(shapes, c["seed"])
    P = {k: (1.0 + 0.1 * v if (("norm" in k) and k.endswith(".weight")) else v) for k, v in st.items()}      # norm scales around 1, as the generator
    assert abs(state_checksum(P) - c["state_checksum"]) < 1e-6 * max(1.0, abs(c["state_checksum"]))       # same names, same shapes, same values
    x = c["x"].clone().requires_grad_(True)
    z = c["z"].clone().requires_grad_(True)
    m = OV.encode_moments(P, cfg, x)
    (m * c["w_m"]).sum().backward()
    px = OV.decode(P, cfg, z)
    (px * c["w_p"]).sum().backward()
    errs = {"moments": rel_l2(m, c["moments"]), "pixels": rel_l2(px, c["pixels"]), "dx": rel_l2(x.grad, c["dx"]), "dz": rel_l2(z.grad, c["dz"])}
    print(f"[ref-pin] vae {case}:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert all(v < 1e-5 for v in errs.values()), errs
