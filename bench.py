#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native diffusion train step.

Workload (BASELINE.json configs[2], per-GPU replica): Flux.1-dev MMDiT (19 double + 38 single blocks, D=3072, 24x128 heads,
guidance embeds), LoRA rank 32 on the attention projections, 1024^2 (latents [B,16,128,128] -> 4096 image tokens + 512 T5
tokens), bf16, AdamW; synthetic latents / text embeddings and random-init weights of the true architecture (no network here).
One "step" = prepare_batch (in-kernel noise + flow noising + target) -> MMDiT forward -> MSE -> hand-written backward ->
(bucketed RCCL all-reduce for N>1) -> fused AdamW, exactly what Trainer.train_step runs.

    python bench.py --gpus N --steps K --warmup W            (N>1: one rank per GPU over RCCL.  Started under torch.distributed.run the
                                                              ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment;
                                                              started bare, bench.py re-executes itself under torch.distributed.run
                                                              --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1)

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel class (the bf16 MFMA
GEMM, measured live with hipEvent pairs on the launch stream by libst355's profiler), `cpu_baseline` (the oracle timed on the
host cores for a bounded sample, N=1 only) and `parity_at_config` (the same sample through the HIP model, compared).  The default
(Flux) line also carries `secondary.sdxl_lora` (the SDXL-LoRA half of BASELINE.json's metric), `secondary.sd3_full_buckets` (configs[3]: SD3-Medium full
fine-tune + EMA over mixed aspect buckets — the full-parameter gradient exchange), `secondary.flux_full_rank` (Flux.1-dev full-rank, AdamWBF16) and
`secondary.sd3_lora_r128_bs3_published_row` (the reference's own published single-GPU row run like for like under hipGraph replay: `published` + `vs_baseline`),
measured after the Flux timing at the same N; at N > 1 every entry carries `comm` (the gradient exchange's per-bucket timings).  `roofline` also names the
MEASURED vendor GEMM ceiling of this pool's boxes (`measured_vendor_gemm_ceiling`, tools/hipblaslt_ceiling.py).  `ms_per_step_stats` =
median / p95 / min / max of the per-step device timestamps; `--model sd15 | sdxl | sd3 --full | pixart` lines carry their own `parity_at_config`.
"""
from __future__ import annotations

import argparse
import datetime
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md chip table); never the 2:1-sparse figure


# the rows of the reference's own benchmark sweep (documentation/experimental/SEGMENTED_CHECKPOINTING.md:795-805, example sd3.peft-lora:
# SD3 LoRA r128 / alpha 128, 1024^2, train_batch_size 3, adamw_bf16, bf16; sec/step post-warm-up on ONE H100) that this bench can run as configured there:
#   python bench.py --model sd3 --rank 128 --batch 3 [--gradient-checkpointing [--ckpt-interval 2 [--ckpt-stride 4]]]
# (the example's optimizer is adamw_bf16 over bf16 adapters: `--optimizer adamw_bf16` trains bf16 adapter values under the fused AdamWBF16 — the default line's
# secondary runs the row that way and carries the fused-fp32-AdamW figure as a second field)
PUBLISHED_SD3_LORA_R128_BS3 = {"none": 0.529, "layer": 0.721, "interval2": 0.723, "seg2-stride4": 0.620}


def published_row(args):
    """-> (mode, H100 sec/step) when the command line is one of the published SD3 rows, else None"""
    if not (args.model == "sd3" and not args.full and int(args.rank) == 128 and int(args.batch) == 3 and args.res == 1024 and args.layers == 19 and not args.buckets):
        return None
    if not args.gradient_checkpointing:
        mode = "none"
    elif args.ckpt_interval in (None, 1) and args.ckpt_stride is None:
        mode = "layer"
    elif args.ckpt_interval == 2 and args.ckpt_stride is None:
        mode = "interval2"
    elif args.ckpt_interval == 2 and args.ckpt_stride == 4:
        mode = "seg2-stride4"
    else:
        return None
    return mode, PUBLISHED_SD3_LORA_R128_BS3[mode]


def train_flops_per_image(n_blocks: int, D: int, S: int) -> float:
    """per block forward = 2*S*12*D^2 (linears) + 4*S^2*D (attention); LoRA train step (no base wgrad, no
    recompute) = 2 x linears (fwd + dgrad) + 3 x attention (fwd + 2x bwd).  Flux.1-dev @1024^2: 1.64e14 FLOP / image."""
    lin = n_blocks * 2.0 * S * 12 * D * D
    att = n_blocks * 4.0 * S * S * D
    return 2 * lin + 3 * att


def hbm_traffic_per_gemm_launch(workload: str = ""):
    """average HBM bytes per GEMM launch from the newest committed PMC summary of this workload (tools/profile_round.sh: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes over the same command, gfx950 correction applied) — or null if none is committed.  workload "" = the default (Flux) command:
    profiles/rNN_hbm_traffic.json; "sdxl_lora": profiles/rNN_sdxl_lora_hbm_traffic.json."""
    import re
    pat = re.compile(r"^r\d+[a-z]*_" + (re.escape(workload) + "_" if workload else "") + r"hbm_traffic\.json$")
    files = sorted(f for f in (ROOT / "profiles").glob("*_hbm_traffic.json") if pat.match(f.name))
    if not files:
        return None, None
    d = json.loads(files[-1].read_text())
    tot, n = 0.0, 0
    for k, v in d["kernels"].items():
        if "k_gemm" in k:
            tot += v["hbm_bytes_per_launch"] * v["launches"]; n += v["launches"]
    return (round(tot / n) if n else None), f"profiles/{files[-1].name}"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="flux", choices=["flux", "sd3", "sdxl", "sd15", "pixart", "vae"],
                    help="flux = the headline workload (BASELINE.json configs[2]); sd3 = SD3-Medium MMDiT LoRA r32 (joint blocks, D=1536), secondary")
    ap.add_argument("--fp8", action="store_true", help="pixart only: fp8-native Linears in the frozen trunk blocks (configs[4]: 'fp8 MFMA')")
    ap.add_argument("--lora", action="store_true", help="sdxl: LoRA on the attention projections instead of the full fine-tune (the metric's SDXL-LoRA); pixart: LoRA on the trunk's attention projections instead of the ControlNet branch")
    ap.add_argument("--graph", action="store_true", help="capture predict + loss + backward into a hipGraph after two eager steps and replay it (launch-bound "
                    "steps: the SDXL UNet); the per-kernel breakdown is then taken from ONE extra eager step after the timed region")
    ap.add_argument("--full", action="store_true", help="flux / sd3 / sdxl / sd15: full-parameter training (every parameter trains, bf16 arena, one fused optimizer launch); sd3 + EMA = BASELINE configs[3], flux = the reference's full-rank datapoint")
    ap.add_argument("--optimizer", default="st355-adamw", choices=["st355-adamw", "adamw_bf16"])
    ap.add_argument("--buckets", action="store_true",
                    help="sd3: cycle the mixed aspect buckets of SURVEY.md §8(d) (latents 128x128, 96x168, 168x96, 112x144, 144x112), one bucket "
                         "per step, each rank starting at a different one (BASELINE configs[3])")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (images).  Defaults: flux 8 (M = 36 864 token rows: 1728 tiles of 256x256 = 6.75 "
                    "waves over 256 CUs, 96 %% wave efficiency; batch 4 is 3.375 waves = 84 %%), sd3 / sdxl / vae 4 (configs[1] names batch 4), sd15 / pixart 1")
    ap.add_argument("--layers", type=int, default=19)
    ap.add_argument("--single-layers", type=int, default=38)
    ap.add_argument("--rank", type=int, default=32)
    ap.add_argument("--lora-target", default="default", choices=["default", "all", "context", "all+ffs", "context+ffs", "all+ffs+embedder", "ai-toolkit", "tiny", "nano"],
                    help="flux only: the reference's flux_lora_target set (flux/model.py:1235-1380); 'default' = attn to_q/to_k/to_v/to_out.0, what BASELINE.json's config names")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--gradient-checkpointing", action="store_true", help="re-run checkpointed blocks in backward instead of keeping their activations "
                    "(SURVEY.md §8(f)3); with --ckpt-interval K [--ckpt-stride S] the reference's segmented modes (interval2 = K 2; seg2-stride4 = K 2 S 4)")
    ap.add_argument("--ckpt-interval", type=int, default=None)
    ap.add_argument("--ckpt-stride", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="flux only: skip the SDXL-LoRA, SD3 full-fine-tune (mixed buckets) and Flux full-rank secondary measurements appended to the default line")
    ap.add_argument("--prof-dump", default=None, help="write one CSV line per launch of the timed steps (class,ms,flops,bytes,shape)")
    ap.add_argument("--no-prof", action="store_true", help="disable the per-launch hipEvent profiler (roofline becomes null)")
    a = ap.parse_args()
    if a.batch is None:
        a.batch = {"flux": 8, "sd3": 8, "sdxl": 4, "vae": 4, "sd15": 1, "pixart": 1}[a.model]     # sd3: 8 (r02: batch 4 -> 8 = 19.7 -> 23.9 images/s full fine-tune)
    return a


_T0 = time.time()


def _log(msg):
    """progress on stderr with the wall clock since process start (the driver keeps stderr; the JSON line stays the only stdout line)"""
    print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


CPU_LEG_BUDGET_S = 75.0          # BOUNDED sample: the timed passes of the host-core leg stop once this much wall clock is spent (the contract asks for 10-30 s of CPU work)


def cpu_baseline(args, dev=None):
    """the oracle (plain-torch restatement of the reference path) on the host cores: 1 double + 1 single block Flux at full width
    (D=3072, 24x128 heads) and full sequence (4096 + 512 tokens), LoRA r32 — forward, autograd backward (fp32) and torch.optim.AdamW over the
    adapters.  One untimed warm-up pass at a short sequence (thread pool, code paths), then up to 3 timed passes at the full shape, stopping early
    once CPU_LEG_BUDGET_S of wall clock is spent (one full-shape pass costs tens of seconds on the host: r02 measured 41 s for the double block);
    each pass times its double-block and single-block halves (forward + backward) and the optimizer step; the MEDIAN of the completed passes is
    extrapolated linearly in block count to the 19+38 stack, the optimizer step in adapter count.  The SAME weights, adapters and
    inputs then go through the HIP model on the device and the outputs are compared: `parity_at_config` = prediction rel-L2 / cosine and the
    worst adapter-gradient rel-L2 at the BASELINE shape (the oracle is the checker here, never the thing measured as the product).
    (2 double + 4 single blocks, as BASELINE.md §3 planned, would be 2-3 minutes per pass on the host cores: outside a default run that must finish
    in minutes — `ST355_CPU_LEG_BLOCKS=2,4` runs it.)"""
    import statistics

    import torch.nn.functional as F

    from oracle import flux as OF

    # `cores` of the JSON = the intra-op threads actually used.  The GPU box shows 256 logical cpus but the container gets a fraction of them: measured r3
    # (tools/probes/cpu_threads_probe.py, profiles/archive/r03_cpu_threads_probe.log) an fp32 4608 x 3072 x 12288 linear runs at 1.51 / 1.68 / 1.30 / 0.95 / 0.47 TFLOP/s on
    # 16 / 32 / 64 / 128 / 256 threads — 32 threads is the fastest this host gets, 256 is 3.6x slower
    cores = int(os.environ.get("ST355_CPU_LEG_THREADS", "0")) or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    ND, NS = (int(v) for v in os.environ.get("ST355_CPU_LEG_BLOCKS", "1,1").split(","))
    cfg = OF.FluxConfig(num_layers=ND, num_single_layers=NS)
    _log(f"cpu_baseline: oracle weights for {ND} double + {NS} single blocks")
    P = OF.init_params(cfg, seed=1)
    P = {k: v.to(torch.bfloat16).float() for k, v in P.items()}           # both sides start from the same bf16-representable weights
    lat = args.res // 8
    S_img, S_txt = (lat // 2) ** 2, 512
    B, r = 1, int(args.rank)
    g = torch.Generator().manual_seed(0)
    bf = lambda t: t.to(torch.bfloat16).float()
    packed = bf(torch.randn(B, S_img, cfg.in_channels, generator=g))
    prompt = bf(torch.randn(B, S_txt, cfg.joint_attention_dim, generator=g))
    pooled = bf(torch.randn(B, cfg.pooled_projection_dim, generator=g))
    tstep = torch.tensor([0.37] * B)
    guidance = torch.full((B,), 1.0)
    lora = OF.init_lora(cfg, P, r, seed=7, b_std=2e-2)
    lora = {k: (bf(a).requires_grad_(True), bf(b).requires_grad_(True)) for k, (a, b) in lora.items()}
    lora0 = {k: (a.detach().clone(), b.detach().clone()) for k, (a, b) in lora.items()}
    img_ids, txt_ids = OF.prepare_latent_image_ids(lat, lat), torch.zeros(S_txt, 3)
    cos, sin = OF.rope_tables(torch.cat([txt_ids, img_ids], 0))
    s_names = [k for k in lora if k.startswith("single_")]
    d_names = [k for k in lora if k.startswith("transformer_blocks")]
    s_par = [t for k in s_names for t in lora[k]]
    d_par = [t for k in d_names for t in lora[k]]
    opt = torch.optim.AdamW(d_par + s_par, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    # embedders + tail are a few GFLOP: outside the timed block regions
    temb = OF.time_text_embed(P, cfg, tstep * 1000, guidance * 1000, pooled).detach()

    def adamw_step():
        """torch.optim.AdamW over the 1.4 M adapter parameters: a handful of tiny element-wise passes, run on 16 threads (with 256 threads every pass paid a
        256-way fork / join: 13 s per step, measured r3)"""
        torch.set_num_threads(min(cores, 16))
        t0 = time.time()
        opt.step()
        dt = time.time() - t0
        torch.set_num_threads(cores)
        return dt

    def one_pass(n_img, step):
        """forward / backward over all blocks on the first n_img image tokens (n_img = S_img: the full shape)"""
        rows = slice(0, n_img)
        c_, s_ = cos[:S_txt + n_img], sin[:S_txt + n_img]
        hid0 = OF.linear(packed[:, rows], P, "x_embedder").detach().requires_grad_(True)
        enc0 = OF.linear(prompt, P, "context_embedder").detach().requires_grad_(True)
        t0 = time.time()
        enc, hid = enc0, hid0
        for i in range(ND):
            enc, hid = OF.double_block(P, cfg, i, hid, enc, temb, c_, s_, lora, 1.0)
        t_fd = time.time() - t0
        x1 = torch.cat([enc, hid], dim=1)
        t0 = time.time()
        x2 = x1
        for i in range(NS):
            x2 = OF.single_block(P, cfg, i, x2, temb, c_, s_, lora, 1.0)
        t_fs = time.time() - t0
        scale_o, shift_o = OF.linear(F.silu(temb), P, "norm_out.linear").chunk(2, dim=1)
        pred = OF.linear(OF.layer_norm(x2[:, S_txt:]) * (1 + scale_o[:, None]) + shift_o[:, None], P, "proj_out")
        loss = pred.float().pow(2).mean()
        (gx2,) = torch.autograd.grad(loss, [x2], retain_graph=True)
        t0 = time.time()
        gs = torch.autograd.grad([x2], [x1] + s_par, grad_outputs=[gx2], retain_graph=True)
        t_bs = time.time() - t0
        t0 = time.time()
        gd = torch.autograd.grad([x1], [hid0, enc0] + d_par, grad_outputs=[gs[0]])
        t_bd = time.time() - t0
        t_opt = 0.0
        if step:
            for t_, g_ in zip(s_par + d_par, list(gs[1:]) + list(gd[2:])):
                t_.grad = g_
            t_opt = adamw_step()
        return (t_fd + t_bd, t_fs + t_bs, t_opt), pred.detach(), (gs[1:], gd[2:])

    _log("cpu_baseline: warm-up pass (256 image tokens)")
    one_pass(256, False)                                                    # warm-up, untimed
    for t_ in s_par + d_par:                                                # ... and one optimizer step on zero gradients: creates the AdamW state, so the timed steps are
        t_.grad = torch.zeros_like(t_)                                      # steady-state steps; then back to the initial adapters (the first timed pass is the one the
    adamw_step()                                                            # device is compared with)
    with torch.no_grad():
        for k, (a, b) in lora.items():
            a.copy_(lora0[k][0]); b.copy_(lora0[k][1])
    # the first timed pass runs from the initial adapters and is the one the device is compared with; AdamW moves the adapters after each pass
    times, pred, keep = [], None, None
    t_leg = time.time()
    for it in range(3):
        tt, pr, kp = one_pass(S_img, True)
        times.append(tt)
        _log(f"cpu_baseline: timed pass {it + 1}: double {tt[0]:.1f} s, single {tt[1]:.1f} s, AdamW {tt[2] * 1e3:.0f} ms")
        if it == 0:
            pred = pr
            keep = {id(t_): g_.detach().clone() for ts, gs_ in ((s_par, kp[0]), (d_par, kp[1])) for t_, g_ in zip(ts, gs_)}
        if (time.time() - t_leg) + sum(tt) > CPU_LEG_BUDGET_S:             # the next pass would overrun the bounded sample
            break
    t_double = statistics.median(t[0] for t in times) / ND
    t_single = statistics.median(t[1] for t in times) / NS
    n_adapter_sample = sum(t.numel() for t in d_par + s_par)
    t_opt = statistics.median(t[2] for t in times) * (args.layers * 4 + args.single_layers * 3) / max(1, ND * 4 + NS * 3)
    step_s = t_double * args.layers + t_single * args.single_layers + t_opt
    ograd = {k: (keep[id(lora[k][0])], keep[id(lora[k][1])]) for k in lora}
    out = {
        "value": round(B / step_s, 6), "unit": "images/s", "cores": cores, "kind": "port",
        "sample": f"oracle (plain torch fp32, autograd, torch.optim.AdamW) {ND} double + {NS} single Flux blocks fwd+bwd+optimizer at D=3072, S={S_img}+{S_txt}, "
                  f"B=1, LoRA r{r} ({n_adapter_sample / 1e6:.1f} M adapter parameters): 1 warm-up pass at a short sequence, median of {len(times)} timed pass(es) "
                  f"(bounded at {CPU_LEG_BUDGET_S:.0f} s of wall clock) = {t_double:.2f} s per double block, {t_single:.2f} s per single block, extrapolated "
                  f"x{args.layers}/x{args.single_layers} blocks + AdamW (16 threads) {t_opt * 1e3:.0f} ms = {step_s:.0f} s/step (per-pass block seconds: {[round(t[0] + t[1], 1) for t in times]})",
    }
    parity = None
    if dev is not None:
        # the same weights / adapters / inputs through the HIP model (C ABI) on the device
        from simpletuner_amd.flux.transformer import FluxTransformer2DModel
        m = FluxTransformer2DModel(num_layers=ND, num_single_layers=NS, guidance_embeds=True, device=dev)
        m.load_flat_state(P)
        m.add_lora_adapter(rank=r, alpha=float(r))
        with torch.no_grad():
            for name, p_ in m.named_parameters():
                if ".lora_A." in name:
                    p_.copy_(lora0[name.split(".lora_A.")[0]][0])
                elif ".lora_B." in name:
                    p_.copy_(lora0[name.split(".lora_B.")[0]][1])
        m.prepare_for_training()
        to = lambda t, dt=torch.bfloat16: t.to(device=dev, dtype=dt)
        hp = m(hidden_states=to(packed), encoder_hidden_states=to(prompt), pooled_projections=to(pooled), timestep=to(tstep, torch.float32),
               img_ids=to(img_ids, torch.float32), txt_ids=to(txt_ids, torch.float32), guidance=to(guidance, torch.float32), return_dict=False)[0]
        hp.float().pow(2).mean().backward()
        torch.cuda.synchronize()
        rel = lambda a, ref: float((a.detach().float().cpu() - ref.detach().float()).norm() / (ref.detach().float().norm() + 1e-30))
        hpf, opf = hp.detach().float().cpu().flatten(), pred.detach().float().flatten()
        worst = (0.0, "")
        for name, p_ in m.named_parameters():
            if ".lora_" in name:
                key = name.split(".lora_")[0]
                worst = max(worst, (rel(p_.grad, ograd[key][0 if ".lora_A." in name else 1]), name))
        parity = {"what": f"{ND} double + {NS} single block Flux (D=3072, 24x128 heads, S={S_img}+{S_txt}, LoRA r{r}): HIP bf16 vs oracle fp32, same weights / inputs",
                  "pred_rel_l2": round(rel(hp, pred), 6), "pred_cos": round(float(torch.dot(hpf, opf) / (hpf.norm() * opf.norm())), 7),
                  "lora_grad_worst_rel_l2": round(worst[0], 6), "lora_grad_worst_at": worst[1], "lora_grads_compared": len(ograd) * 2,
                  "tolerance": "pred rel_l2 <= 2e-2, cos >= 0.9995, adapter grads rel_l2 <= 5e-2 (DESIGN.md §3; the oracle itself reproduces the executed reference model class to <= 1e-5, tests/golden/ref_flux_model.pt)"}
        del m
    return out, parity


def cpu_baseline_unet(args, sd15: bool, lora: bool):
    """BASELINE.json configs[0] literally ("SD 1.5 UNet LoRA rank=16, 512^2, batch=1, CPU reference trainer") and the SDXL analogue: ONE full train
    step of the oracle restatement on the host cores — UNet forward (fp32), epsilon MSE, autograd backward, torch.optim.AdamW — at the bench's
    resolution, batch 1, true architecture (859.5 M / 2.57 B parameters), random-init weights.  LoRA: peft-style adapters on every attn1 / attn2
    to_q, to_k, to_v, to_out.0, applied as merged weights W + (alpha/r) B A with autograd through the merge (adapter gradients only)."""
    from oracle.unet import UNetConfig, init_params, unet_forward

    cores = int(os.environ.get("ST355_CPU_LEG_THREADS", "0")) or min(os.cpu_count() or 1, 32)          # see cpu_baseline: 32 threads is this host's fastest
    torch.set_num_threads(cores)
    cfg = UNetConfig.sd15() if sd15 else UNetConfig()
    P = init_params(cfg, seed=1)
    lat, B, r = args.res // 8, 1, int(args.rank)
    g = torch.Generator().manual_seed(0)
    x, noise = torch.randn(B, 4, lat, lat, generator=g), torch.randn(B, 4, lat, lat, generator=g)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    ack = None
    if cfg.addition_embed_type == "text_time":
        ack = {"text_embeds": torch.randn(B, 1280, generator=g), "time_ids": torch.tensor([[args.res, args.res, 0, 0, args.res, args.res]] * B, dtype=torch.float32)}
    t = torch.tensor([500.0] * B)
    if lora:
        targets = [k[:-len(".weight")] for k in P if k.endswith((".to_q.weight", ".to_k.weight", ".to_v.weight", ".to_out.0.weight"))]
        A = {n: (torch.randn(r, P[n + ".weight"].shape[1], generator=g) / math.sqrt(P[n + ".weight"].shape[1])).requires_grad_(True) for n in targets}
        Bm = {n: (1e-3 * torch.randn(P[n + ".weight"].shape[0], r, generator=g)).requires_grad_(True) for n in targets}
        train = list(A.values()) + list(Bm.values())
    else:
        train = [v.requires_grad_(True) for v in P.values()]
    opt = torch.optim.AdamW(train, lr=1e-4 if lora else 1e-5)
    t0 = time.time()
    Pe = dict(P)
    if lora:
        for n in targets:
            Pe[n + ".weight"] = P[n + ".weight"] + Bm[n] @ A[n]          # alpha = r
    pred = unet_forward(Pe, cfg, x, t, ctx, ack)
    loss = ((pred - noise) ** 2).mean()
    loss.backward()
    opt.step()
    step_s = time.time() - t0
    what = f"LoRA r{r} on {len(targets)} attention projections" if lora else "full fine-tune"
    return {
        "value": round(B / step_s, 6), "unit": "images/s", "cores": cores, "kind": "port",
        "sample": f"oracle (plain torch fp32, autograd, torch.optim.AdamW) ONE full {'SD 1.5' if sd15 else 'SDXL'} UNet train step, {what}, {args.res}^2, batch 1: "
                  f"{step_s:.1f} s (loss {float(loss.detach()):.4f})",
    }


def bench_vae(args, dev, rank, world):
    """secondary workload: the VAE latent encode of the hot path — AutoencoderKL.encode + sample + scale of a
    [B,3,res,res] batch with the SDXL VAE architecture (128/256/512/512 channels), random-init weights; a 'step' = one batch encode"""
    import torch.distributed as dist
    from simpletuner_amd import ops
    from simpletuner_amd.vae.autoencoder_kl import AutoencoderKL
    from tools.flop_count import vae_encoder_flops as encoder_flops      # arithmetic over the model's own config
    vae = AutoencoderKL(device=dev)
    vae.load_state_dict(vae.synthetic_state_dict(42))
    B = args.batch
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    x = (torch.rand(B, 3, args.res, args.res, device=dev, generator=gen) * 2 - 1).to(torch.bfloat16)
    for _ in range(args.warmup):
        z = vae.encode_scaled(x, generator=gen)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if not args.no_prof:
        ops.prof_reset(); ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        z = vae.encode_scaled(x, generator=gen)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = None
    if not args.no_prof:
        ops.prof_enable(False); prof = ops.prof_collect(); ops.prof_reset()
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        fl = encoder_flops(vae.config, args.res, args.res) * B
        ms = elapsed / args.steps * 1e3
        roof = kernels = None
        if prof is not None and prof["gemm"]["ms"] > 0:
            g = prof["gemm"]
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "k_gemm_* (conv-as-GEMM)", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None, "launches_per_step": g["launches"] // args.steps,
                    "share_of_step": round(g["ms"] / (elapsed * 1e3), 3)}
            kernels = {k: {"ms_per_step": round(v["ms"] / args.steps, 2), "launches_per_step": v["launches"] // args.steps} for k, v in prof.items() if v["launches"]}
        print(json.dumps({"metric": f"VAE latent encode images/sec (whole node), AutoencoderKL (SDXL VAE architecture) {args.res}^2", "value": round(world * B * args.steps / elapsed, 3),
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": f"AutoencoderKL.encode + sample + scale, [B,3,{args.res},{args.res}] -> [B,4,{args.res // 8},{args.res // 8}], 128/256/512/512 ch, random-init weights",
                                     "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}"},
                          "step_model_tflops": round(fl / (ms * 1e-3) / 1e12, 1), "latent_std": round(float(z.float().std()), 4), "roofline": roof, "kernels": kernels,
                          "cpu_baseline": None}))
    if world > 1:
        dist.destroy_process_group()


def _spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with no launcher in the environment: re-exec under torch.distributed.run, one rank per GPU of this node
    (what the driver's own `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` command does), and hand
    back its exit code.  Rank 0 of the child job prints the JSON line."""
    import socket
    import subprocess
    share = os.environ.get("ST355_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node (one rank per GPU; set ST355_BENCH_SHARE_GPU=1 for the "
                         f"gloo plumbing run that puts every rank on cuda:0)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: spawning {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path for the product")
    if os.environ.get("ST355_BENCH_SHARE_GPU") == "1":     # plumbing test only: all ranks on cuda:0 over gloo (a 1-GPU box cannot run RCCL)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("ST355_COMM_TIMING", "1")      # GradSync brackets every bucket's collective with events on the comm stream: `comm` on the JSON line
        if os.environ.get("ST355_BENCH_SHARE_GPU") == "1":
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
        else:
            # nccl == RCCL over xGMI on ROCm; a collective that a peer never joins aborts after 10 minutes (the watchdog raises on every rank) instead of waiting forever
            dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=600))
    single_rank_pg = world == 1 and os.environ.get("ST355_BENCH_SINGLE_RANK_PG") == "1"
    if single_rank_pg:
        # lab hook for a 1-GPU box: a torch.distributed nccl (= RCCL) group of ONE rank, with GradSync issuing every bucket's collectives anyway
        # (ST355_COMM_SINGLE_RANK): the overlapped exchange of the real step at its real arena size runs over RCCL — stream ordering, work handles, the
        # comm-stream join — and `comm` on the line gives the collectives' own device time (no peer, so no wire time); the step's numbers must not move
        import tempfile
        os.environ["ST355_COMM_SINGLE_RANK"] = "1"
        os.environ.setdefault("ST355_COMM_TIMING", "1")
        dist.init_process_group(backend="nccl", init_method=f"file://{tempfile.mkdtemp()}/pg", rank=0, world_size=1, device_id=dev,
                                timeout=datetime.timedelta(seconds=600))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree (n_gpus in the JSON line is the rank count)")
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        if rank == 0:
            print(f"[bench] {dist.get_backend()} process group up: {dist.get_world_size()} ranks (one per GPU), rank 0 on {torch.cuda.get_device_name(dev)}", file=sys.stderr)

    if args.model == "vae":
        return bench_vae(args, dev, rank, world)
    out = run_workload(args, dev, rank, world)
    if rank == 0:
        _log(f"{args.model}: workload done")
    if args.model == "flux" and not args.no_secondary:
        # the metric names "SDXL-LoRA & Flux-dev 1024^2": the SDXL-LoRA half rides along as a secondary measurement of the same run
        # (r16 on attn1/attn2 to_q/to_k/to_v/to_out.0, a few steps; hipGraph replay at every N — the gradient exchange follows each replay).  Per-GPU batch 32 (r6; 16
        # until r5, kept as the `same_tree_at_batch_16` field): the UNet's 32^2 / 64^2 levels give a 256-CU chip too few tiles per launch at small batches (measured
        # r02, same box: batch 4 / 8 / 16 = 23.8 / 29.9 / 37.3 images/s; r06, same box and tree: batch 16 / 24 / 32 / 36 = 41.3 / 43.7 / 45.1 / 44.8 images/s) and the
        # captured step of batch 32 peaks at 180 GiB of the 288; `--model sdxl --lora --rank 16 --batch 4` is configs[1]'s batch
        import copy
        import gc
        keys = ("metric", "value", "unit", "ms_per_step", "ms_per_step_stats", "steps", "warmup", "config", "step_model_tflops", "step_frac_of_bf16_mfma_peak",
                "roofline", "loss", "peak_hbm_gib", "published_context", "comm")
        a2 = copy.copy(args)
        a2.model, a2.lora, a2.rank, a2.batch, a2.full, a2.graph, a2.buckets = "sdxl", True, 16, 32, False, True, False
        # BASELINE.json configs[3]: SD3-Medium full fine-tune + EMA over the mixed aspect buckets — the full-parameter gradient exchange (2.0 B bf16 gradients
        # per step, reduce-scatter + all-gather buckets behind the backward) and the shared token-balanced bucket schedule run at every N the driver launches
        a3 = copy.copy(args)
        # (eager at every N: one captured step per aspect bucket — `--buckets --graph` — fits since r6 (every capture on ONE side stream + one shared pool: 84 GiB
        # peak for the five bucket graphs) but buys nothing: 302.5 ms under replay against ~300 eager on the same tree — this step is not launch-bound)
        # per-GPU batch 16 (r6; 8 until r5, kept as `same_tree_at_batch_8`): the fused optimizer + EMA pass (9 ms over 2.0 B parameters) and the split-K weight gradients
        # amortise over twice the tokens — r06, same box and tree: batch 8 / 12 / 16 / 24 / 32 = 26.7 / 28.3 / 28.8 / 29.4 / 29.9 images/s at 84 / 108 / 133 / 183 / 233 GiB
        a3.model, a3.lora, a3.rank, a3.batch, a3.full, a3.graph, a3.buckets = "sd3", False, 32, 16, True, False, True
        # Flux.1-dev FULL-rank (11.9 B bf16 parameters, AdamWBF16, per-GPU batch 8 — the configuration of the reference's multi-GPU Flux datapoint,
        # documentation/DISTRIBUTED.md:291-298): hand-written backward with every weight / bias / modulation / norm gradient, one fused optimizer launch over the
        # parameter arena, and at N > 1 the whole 24 GB bf16 gradient arena exchanged per step (fp32-accumulating reduce-scatter + all-gather buckets behind the
        # backward).  Checkpoint plan interval 3 / stride 4: three of every four blocks are recomputed, the fourth keeps its activations (peak 211 GiB of 268)
        a4 = copy.copy(args)
        a4.model, a4.lora, a4.rank, a4.batch, a4.full, a4.graph, a4.buckets = "flux", False, 32, 8, True, False, False
        # The reference's own published single-GPU row that this path covers like for like (example sd3.peft-lora — SD3-Medium LoRA r128, 1024^2,
        # batch 3, bf16, no activation checkpointing: 0.529 s/step on one H100): hipGraph replay of the captured step; `published` + `vs_baseline` ride on its entry
        a5 = copy.copy(args)
        a5.model, a5.lora, a5.rank, a5.batch, a5.full, a5.graph, a5.buckets, a5.res = "sd3", False, 128, 3, False, True, False, 1024
        if rank == 0:
            out["secondary"] = {}
        keys = keys + ("published", "vs_baseline")
        for name, a_ in (("sdxl_lora", a2), ("sd3_full_buckets", a3), ("flux_full_rank", a4), ("sd3_lora_r128_bs3_published_row", a5)):
            a_.steps, a_.warmup, a_.no_cpu_baseline, a_.prof_dump, a_.fp8, a_.gradient_checkpointing = min(args.steps, 5), 2, True, None, False, False
            if name == "flux_full_rank":
                a_.steps, a_.warmup, a_.optimizer, a_.gradient_checkpointing, a_.ckpt_interval, a_.ckpt_stride = min(args.steps, 3), 1, "adamw_bf16", True, 3, 4
            if name == "sd3_lora_r128_bs3_published_row":
                a_.steps, a_.warmup = max(min(args.steps, 8), 5), 3          # two eager steps precede the capture
                a_.optimizer = "adamw_bf16"                                  # the example's optimizer (simpletuner/examples/sd3.peft-lora/config.json): like for like
            gc.collect()
            torch.cuda.empty_cache()              # the previous workload's cached blocks go back before the next one's pools are built
            try:                                  # a secondary must never take the headline line down with it (every rank runs the same code: they fail together)
                if rank == 0:
                    _log(f"secondary {name}")
                sec = run_workload(a_, dev, rank, world)
                if rank == 0:
                    out["secondary"][name] = {k: sec[k] for k in keys if k in sec}
                del sec
                if name in ("sdxl_lora", "sd3_full_buckets"):              # second field: the batch of rounds 2-5, same tree and box (round-over-round comparison)
                    a7 = copy.copy(a_)
                    a7.batch = 16 if name == "sdxl_lora" else 8
                    gc.collect(); torch.cuda.empty_cache()
                    sec = run_workload(a7, dev, rank, world)
                    if rank == 0:
                        out["secondary"][name][f"same_tree_at_batch_{a7.batch}"] = {k: sec[k] for k in ("value", "unit", "ms_per_step", "step_frac_of_bf16_mfma_peak", "loss", "peak_hbm_gib", "roofline") if k in sec}
                    del sec
                if name == "sd3_lora_r128_bs3_published_row":             # second field: the same row under the fused fp32 AdamW over the fp32 adapter arena
                    a6 = copy.copy(a_)
                    a6.optimizer = "st355-adamw"
                    gc.collect(); torch.cuda.empty_cache()
                    sec = run_workload(a6, dev, rank, world)
                    if rank == 0:
                        out["secondary"][name]["same_row_under_fp32_adamw"] = {k: sec[k] for k in ("value", "unit", "ms_per_step", "vs_baseline", "loss") if k in sec}
                    del sec
            except Exception as e:                # noqa: BLE001
                if rank == 0:
                    out["secondary"][name] = {"error": f"{type(e).__name__}: {e}"[:400]}
        if world == 1 and not args.no_cpu_baseline and rank == 0:
            # the oracle as the CHECKER of the HIP components at the other four BASELINE.json configurations, all at their true widths, sequence lengths AND
            # depths (SD3-Medium 24 joint blocks on one of its mixed aspect buckets, PixArt-Sigma 28 trunk + 13 ControlNet blocks at 2K) —
            # tests/parity_at_config.py, the same functions the GPU tests assert on
            from tests import parity_at_config as PC
            others = {}
            for name, fn in (("configs[0] sd15_lora_r16_512", lambda: PC.unet("sd15", 512, dev, lora=True, rank=16)),
                             ("configs[1] sdxl_lora_r16_1024", lambda: PC.unet("sdxl", 1024, dev, lora=True, rank=16)),
                             ("configs[3] sd3_full_finetune_full_depth_bucket_1216x832", lambda: PC.sd3_full(1024, dev, layers=24, hw=(1216, 832))),
                             ("configs[4] pixart_controlnet_2k_full_depth", lambda: PC.pixart_controlnet(2048, dev, trunk_layers=28, ctrl_layers=13))):
                gc.collect()
                torch.cuda.empty_cache()
                if time.time() - _T0 > 480.0:     # the default run must finish in minutes: what did not fit is named, not silently dropped
                    others[name] = {"skipped": "time budget of the default run (the GPU suite asserts it: tests/test_parity_at_config_gpu.py)"}
                    continue
                try:
                    _log(f"parity at {name}")
                    t0 = time.time()
                    others[name] = fn()
                    others[name]["seconds"] = round(time.time() - t0, 1)
                except Exception as e:            # noqa: BLE001
                    others[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
            out["parity_at_other_configs"] = others
    if rank == 0:
        # RCCL writes its version banner through C stdio: with stdout redirected to a file or pipe those bytes sit in libc's buffer until exit and would land AFTER
        # the JSON line (seen on the MI355X, r06) — flush them out first so that the JSON line is the last line of stdout at every N
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:               # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def run_workload(args, dev, rank, world):
    """build one workload, run warmup + timed steps, return the JSON dict on rank 0 (None elsewhere)"""
    torch.cuda.reset_peak_memory_stats(dev)          # `peak_hbm_gib` is per workload
    from simpletuner_amd import ops
    from simpletuner_amd.training.trainer import St355Accelerator, Trainer, default_config

    cfg = default_config(model_family=args.model, lora_rank=args.rank, train_batch_size=args.batch, seed=42, lora_init_b_std=1e-3,   # weights / adapter init / rounding seeds are REPLICA-identical; the data RNG below is per rank
                         flux_lora_target=getattr(args, "lora_target", "default"),
                        
                         model_type="full" if args.full else "lora", use_ema=bool(args.full) and args.model != "flux", optimizer=args.optimizer,
                         learning_rate=1e-5 if args.full else 1e-4, hip_graph=bool(args.graph),
                         gradient_checkpointing=bool(getattr(args, "gradient_checkpointing", False)),
                         gradient_checkpointing_interval=getattr(args, "ckpt_interval", None), gradient_checkpointing_segment_stride=getattr(args, "ckpt_stride", None))
    acc = St355Accelerator(dev)
    if args.model == "flux":
        from simpletuner_amd.flux.model import Flux
        plugin = Flux(cfg, acc)
        plugin.load_model(num_layers=args.layers, num_single_layers=args.single_layers, guidance_embeds=True)
        n_blocks, D_model, S_txt, txt_dim, pooled_dim = args.layers + args.single_layers, 3072, 512, 4096, 768
        tset = getattr(args, "lora_target", "default")
        desc = (f"Flux.1-dev MMDiT ({args.layers} double + {args.single_layers} single, D=3072, 24x128 heads) LoRA r{args.rank} "
                f"on {'attn to_q/to_k/to_v/to_out.0' if tset == 'default' else 'flux_lora_target=' + tset}, {args.res}^2 (S=4096+512), AdamW, random-init weights")
    elif args.model == "sdxl":
        # BASELINE.json configs[1]: SDXL UNet full fine-tune bf16, 1024^2 bucket, batch 4; --lora = the metric's SDXL-LoRA (adapters on attn1/attn2)
        from simpletuner_amd.sdxl.model import SDXL
        from tools.flop_count import unet_flops_fwd            # arithmetic over the model's own config
        sdxl_lora = bool(args.lora)
        args.full = not sdxl_lora
        cfg.model_type, cfg.use_ema, cfg.learning_rate = ("lora" if sdxl_lora else "full"), False, (1e-4 if sdxl_lora else 1e-5)
        plugin = SDXL(cfg, acc)
        plugin.load_model()
        S_txt, txt_dim, pooled_dim = 77, 2048, 1280
        n_blocks, D_model = 0, 0
        sdxl_fwd_flops = unet_flops_fwd(plugin.model.config, args.res // 8, args.res // 8, 77)
        desc = (f"SDXL UNet2DConditionModel (320/640/1280 ch, 2/10-layer transformers at 64^2/32^2, 2.6 B params) "
                f"{f'LoRA r{args.rank} on attn1/attn2 to_q/to_k/to_v/to_out.0' if sdxl_lora else 'FULL fine-tune bf16'}, "
                f"{args.res}^2 ({args.res // 8}^2 latents), epsilon objective, AdamW, random-init weights")
    elif args.model == "sd15":
        # BASELINE.json configs[0]: SD 1.5 UNet LoRA rank 16, 512^2, batch 1 (the reference's CPU-runnable plumbing case) — run it with
        # `--model sd15 --rank 16 --res 512 --batch 1`; --full switches to the full fine-tune
        from simpletuner_amd.sd1x.model import StableDiffusion1
        from tools.flop_count import unet_flops_fwd
        sd15_lora = not args.full
        cfg.model_type, cfg.use_ema, cfg.learning_rate = ("lora" if sd15_lora else "full"), False, (1e-4 if sd15_lora else 1e-5)
        plugin = StableDiffusion1(cfg, acc)
        plugin.load_model()
        S_txt, txt_dim, pooled_dim = 77, 768, 0
        n_blocks, D_model = 0, 0
        sdxl_fwd_flops = unet_flops_fwd(plugin.model.config, args.res // 8, args.res // 8, 77)
        desc = (f"SD 1.5 UNet2DConditionModel (320/640/1280/1280 ch, 8 heads of width 40/80/160, 0.86 B params) "
                f"{f'LoRA r{args.rank} on attn1/attn2 to_q/to_k/to_v/to_out.0' if sd15_lora else 'FULL fine-tune bf16'}, {args.res}^2 "
                f"({args.res // 8}^2 latents), epsilon objective, AdamW, random-init weights")
    elif args.model == "pixart":
        # BASELINE.json configs[4]: PixArt-Sigma DiT, ControlNet branch (13 copied blocks) trained, 2K latents (256^2 x 4), T5 ctx 300 with mask
        from simpletuner_amd.pixart.model import PixartSigma
        from tools.flop_count import pixart_flops_fwd
        pix_lora = bool(args.lora)          # --lora: a LoRA on the trunk's attention projections (pixart/model.py:59) instead of the ControlNet branch
        cfg.model_type, cfg.use_ema, cfg.learning_rate = ("lora" if pix_lora else "full"), False, (1e-4 if pix_lora else 1e-5)
        plugin = PixartSigma(cfg, acc)
        plugin.load_model(sample_size=256 if args.res >= 2048 else 128, fp8_base=bool(args.fp8))
        if pix_lora:
            plugin.add_lora_adapter()
        else:
            plugin.controlnet_init(num_layers=13, synthetic_adapter=True)
        S_txt, txt_dim, pooled_dim = 300, 4096, 0
        n_blocks, D_model = 0, 0
        lat_ = args.res // 8
        f_trunk = pixart_flops_fwd(plugin.model.config, lat_, lat_, 300, 0)
        f_blk = f_trunk / 28.0
        # trunk: forward (28) + input gradients through blocks 1..27 (2x forward each: attention bwd = 2x fwd, linears dgrad = 1x -> ~1.7x; counted 1x
        # for the linears and 2x for attention is folded into 2x here);  adapter (13): forward + dgrad + wgrad = 3x.  LoRA on the trunk: forward + input gradients
        # through all 28 blocks (the rank-space products are < 1 % of them)
        pix_step_flops = (f_trunk + 2.0 * 28 * f_blk) if pix_lora else (f_trunk + 2.0 * 27 * f_blk + 3.0 * 13 * f_blk)
        desc = ((f"PixArt-Sigma XL/2 (28 blocks, 16x72 heads, D=1152) LoRA r{args.rank} on attn1 / attn2 to_q/to_k/to_v/to_out.0 of every block, " if pix_lora else
                 f"PixArt-Sigma XL/2 (28 blocks, 16x72 heads, D=1152) ControlNet-Transformer branch (13 copied blocks + zero-init projections) trained, trunk "
                 f"frozen{' with fp8-native Linears (e5m2 x e4m3 MFMA)' if args.fp8 else ''}, ")
                + f"{args.res}^2 ({lat_}^2 latents, S={(lat_ // 2) ** 2}), T5 ctx 300 (120 valid), epsilon objective, AdamW, random-init weights")
    else:
        from simpletuner_amd.sd3.model import SD3
        plugin = SD3(cfg, acc)
        n_l = 24 if args.layers == 19 else args.layers          # SD3-Medium: 24 joint blocks, 24 x 64 heads
        plugin.load_model(sample_size=128, num_layers=n_l, num_attention_heads=24, attention_head_dim=64, caption_projection_dim=1536,
                          pooled_projection_dim=2048, pos_embed_max_size=192)
        n_blocks, D_model, S_txt, txt_dim, pooled_dim = n_l, 1536, 231, 4096, 2048
        desc = (f"SD3-Medium MMDiT ({n_l} joint blocks, D=1536, 24x64 heads) LoRA r{args.rank} on attn to_q/to_k/to_v/to_out.0, "
                f"{args.res}^2 (S=4096+231), AdamW, random-init weights")
    if args.model == "pixart":
        pass
    elif args.full:
        if args.model not in ("sd3", "sdxl", "sd15", "flux"):
            raise SystemExit("--full is wired for --model flux / sd3 / sdxl / sd15")
        plugin.enable_full_finetune()
        comp_ = plugin.get_trained_component()
        n_par = sum(p.numel() for p in (comp_.trainable_parameters() if hasattr(comp_, "trainable_parameters") else [q for q in comp_.parameters() if q.requires_grad]))
        desc = desc.replace(f"LoRA r{args.rank} on attn to_q/to_k/to_v/to_out.0", f"FULL-rank training ({n_par / 1e9:.1f} B bf16 params){' + EMA' if cfg.use_ema else ''}")
    elif args.model != "pixart":
        plugin.add_lora_adapter()
    trainer = Trainer(cfg, plugin, acc)

    lat = args.res // 8
    B = args.batch
    torch.manual_seed(42 + rank)              # sigma sampling draws from the global device generator (examples' `seed: 42`)
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    def make_batch(hh=None, ww=None):
        hh, ww = hh or lat, ww or lat
        b = {
            "latent_batch": torch.randn(B, 4 if args.model in ("sdxl", "pixart", "sd15") else 16, hh, ww, device=dev, generator=gen).to(torch.bfloat16),
            "prompt_embeds": torch.randn(B, S_txt, txt_dim, device=dev, generator=gen).to(torch.bfloat16),
            "add_text_embeds": torch.randn(B, max(pooled_dim, 8), device=dev, generator=gen).to(torch.bfloat16),
        }
        if args.model == "sdxl":       # time_ids [B,6] = (1024,1024,0,0,1024,1024)
            b["batch_time_ids"] = torch.tensor([[args.res, args.res, 0, 0, args.res, args.res]] * B, device=dev, dtype=torch.bfloat16)
        if args.model == "pixart":     # ctx [B,300,4096] with mask (first 120 valid), conditioning_latents of the latent shape
            del b["add_text_embeds"]
            m_ = torch.zeros(B, S_txt, device=dev, dtype=torch.bfloat16); m_[:, :120] = 1
            b["encoder_attention_mask"] = m_
            if not args.lora:
                b["conditioning_latents"] = torch.randn(B, 4, hh, ww, device=dev, generator=gen).to(torch.bfloat16)
        return b
    if args.buckets:
        if args.model != "sd3":
            raise SystemExit("--buckets is wired for --model sd3 only (Flux bench keeps the single 1024^2 bucket of configs[2])")
        # the replicas walk ONE shared, seeded bucket schedule (training/bucket_split.py): every step runs the same bucket — the same token count — on
        # every rank, each rank drawing from its own slice of that bucket (the reference lets ranks draw buckets independently, sampler.py:1041-1146)
        from simpletuner_amd.training.bucket_split import TokenBalancedSchedule, split_buckets_between_processes
        shape_of = {"128x128": (128, 128), "96x168": (96, 168), "168x96": (168, 96), "112x144": (112, 144), "144x112": (144, 112)}
        n_sched = max(args.steps, args.warmup)
        per_bucket = (-(-n_sched // len(shape_of)) + 1) * B * world
        local = split_buckets_between_processes({k: [f"{k}/{i}" for i in range(per_bucket)] for k in shape_of}, B, world, rank, seed=42, backend_id="bench")
        sched = TokenBalancedSchedule(local, B, seed=42, epoch=0, tokens_of={k: (h_ // 2) * (w_ // 2) + S_txt for k, (h_, w_) in shape_of.items()})
        by_shape = {k: make_batch(*hw) for k, hw in shape_of.items()}
        batches = [by_shape[b_] for b_ in sched.order[:n_sched]]
        desc += "; mixed aspect buckets " + ",".join(sorted(shape_of)) + " (latent), one shared token-balanced schedule over all ranks"
    else:
        batches = [make_batch() for _ in range(2)]   # resident in HBM before the timed region
    nb_ = len(batches)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trace_loss = os.environ.get("ST355_BENCH_TRACE_LOSS") == "1"      # debugging aid: per-step loss (forces a host sync per step)
    if rank == 0:
        _log(f"{args.model}: model + batches built, {args.warmup} warm-up + {args.steps} timed steps")
    if args.graph and args.buckets:            # one capture per bucket shape before the warm-up proper (each first encounter = 2 eager steps + capture + replay), untimed
        for k_ in sorted(by_shape, key=lambda k__: -shape_of[k__][0] * shape_of[k__][1]):      # largest token count first: the later, smaller captures reuse its pool blocks
            trainer.train_step(dict(by_shape[k_]))
    if os.environ.get("ST355_BENCH_FAIL_RANK") == str(rank) and world > 1:      # lab hook (tools/gpu_lease.sh two_ranks): this rank dies before its first step — the job must end non-zero, promptly
        raise RuntimeError(f"ST355_BENCH_FAIL_RANK: rank {rank} fails on purpose")
    for i in range(args.warmup):
        l_ = trainer.train_step(dict(batches[i % nb_]))
        if trace_loss:
            print(f"[bench] warmup {i} loss {float(l_):.5f}", file=sys.stderr)
    sync()
    prof_live = (not args.no_prof) and not args.graph        # hipGraph replays run no host code: the per-launch events are taken from one eager step below
    if prof_live:
        ops.prof_reset(); ops.prof_enable(True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # per-step device timestamps on the stream the step's kernels run on
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = trainer.train_step(dict(batches[i % nb_]))
        marks[i + 1].record()
        if trace_loss:
            comp_ = plugin.get_trained_component()
            gf_ = getattr(comp_, "_last_grad_flat", None)
            print(f"[bench] step {i} loss {float(loss):.5f} grad_nan {bool(torch.isnan(gf_.float()).any()) if gf_ is not None else None} "
                  f"w_nan {bool(torch.isnan(comp_.arena.float()).any()) if hasattr(comp_, 'arena') else None}", file=sys.stderr)
    sync()
    elapsed = time.perf_counter() - t0
    per_step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    step_stats = {"median": round(per_step_ms[len(per_step_ms) // 2], 2), "p95": round(per_step_ms[min(len(per_step_ms) - 1, int(0.95 * len(per_step_ms)))], 2),
                  "min": round(per_step_ms[0], 2), "max": round(per_step_ms[-1], 2), "source": "device timestamps (hipEvents on the compute stream) per step, this rank"}
    prof = None
    prof_steps = args.steps
    if args.graph and not args.no_prof:
        trainer._use_graph = False
        trainer.optimizer.zero_grad(set_to_none=True)
        trainer.release_graphs()                 # the captured step's pool (173 GiB at SDXL-LoRA batch 32) goes back before the eager step needs its own activations
        ops.prof_reset(); ops.prof_enable(True)
        trainer.train_step(dict(batches[0]))
        sync()
        prof_steps = 1
    if not args.no_prof:
        ops.prof_enable(False)
        prof = ops.prof_collect()
        if args.prof_dump and rank == 0:
            ops.prof_dump(args.prof_dump)
        ops.prof_reset()
    comm_rep = None
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        gs_ = getattr(plugin.get_trained_component(), "grad_sync", None)
        rep_ = gs_.overlap_report() if gs_ is not None else None      # the LAST timed step's exchange (device timestamps; synchronises)
        if rep_ is not None:
            durs = [round((s_["end_ms"] - s_["start_ms"]) * 1e3, 1) for s_ in rep_["slices"]]
            comm_rep = {"path": "st355_comm_* C ABI over RCCL (ST355_COMM=native)" if gs_.comm is not None else (f"torch.distributed {dist.get_backend()} collectives on a comm stream" + (" (nccl = RCCL over xGMI)" if str(dist.get_backend()).lower() == "nccl" else " (gloo: the shared-GPU plumbing run)")),
                        "ranks": int(dist.get_world_size()), "mode": gs_.mode, "fp32_reduce": bool(gs_.fp32_reduce), "bucket_bytes": int(gs_.bucket_elems * gs_.flat.element_size()),
                        "arena_bytes": int(gs_.flat.numel() * gs_.flat.element_size()), "backward_ms": round(rep_["backward_ms"], 3),
                        "comm_ms_sum_over_buckets": round(rep_["comm_ms"], 3), "exposed_tail_ms": round(rep_["exposed_ms"], 3),
                        "overlap_frac": round(rep_["overlap_frac"], 4), "buckets": len(durs), "bucket_us": durs[:64],
                        "what": "last timed step, rank 0: per-bucket collective durations on the comm stream, comm time left after the last backward kernel"}
        elif gs_ is None:
            comm_rep = {"path": "gradient exchange after the step (hipGraph replay or gradient accumulation): no overlapped GradSync on this workload"}
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    loss_val = float(loss.item())

    if rank == 0:
        S_img = (lat // 2) ** 2
        step_flops = train_flops_per_image(n_blocks, D_model, S_img + S_txt) * B
        if args.model == "flux" and not args.full and getattr(args, "lora_target", "default") in ("tiny", "nano"):
            # single_transformer_blocks.7(.20).proj_out only: the backward runs from the last block down to single block 7 and stops — forward over every block, backward
            # (1x linears, 2x attention) over the blocks at or above it
            S_ = S_img + S_txt
            per_lin, per_att = 2.0 * S_ * 12 * D_model * D_model, 4.0 * S_ * S_ * D_model
            n_bwd = max(0, args.single_layers - 7)
            step_flops = (n_blocks * (per_lin + per_att) + n_bwd * (per_lin + 2 * per_att)) * B
        if args.model == "pixart":
            step_flops = pix_step_flops * B
        elif args.model in ("sdxl", "sd15") and not args.full:   # LoRA: forward + input gradients (no base weight gradients): 2x forward, attention bwd 2x
            step_flops = 2.0 * sdxl_fwd_flops * B
        elif args.model in ("sdxl", "sd15"):   # full fine-tune = 3x the forward (fwd + dgrad + wgrad; attention fwd + 2x bwd), tools/flop_count.py::unet_flops_fwd
            step_flops = 3.0 * sdxl_fwd_flops * B
        elif args.full:      # full fine-tune: fwd + dgrad + wgrad on the linears (3x), attention fwd + 2x bwd (3x)  
            step_flops = 3.0 * (n_blocks * 2.0 * (S_img + S_txt) * 12 * D_model * D_model + n_blocks * 4.0 * (S_img + S_txt) ** 2 * D_model) * B
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        roof = None
        kernels = None
        if prof is not None:
            g = prof["gemm"]
            if g["ms"] > 0:
                ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
                # the committed PMC passes: of the default command, and of the metric's second workload (SDXL-LoRA, eager launches of the same kernels)
                traffic, tsrc = (hbm_traffic_per_gemm_launch() if (args.model == "flux" and not args.full) else
                                 hbm_traffic_per_gemm_launch("sdxl_lora") if (args.model == "sdxl" and args.lora and B == 16) else (None, None))     # (that counter pass is of batch 16; at batch 32 rocprofv3 --pmc aborts with an AQL packet error, r06)
                roof = {"bound": "mfma", "kernel": "k_gemm_* (all schedules / epilogues)", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes / launch",
                        "traffic_source": tsrc, "algorithmic_bytes_per_launch": round(g["bytes"] / max(1, g["launches"])),
                        "launches_per_step": g["launches"] // prof_steps, "avg_launch_us": round(g["ms"] * 1e3 / max(1, g["launches"]), 1),
                        "share_of_step": round(g["ms"] / prof_steps / (elapsed / args.steps * 1e3), 3),
                        # the ceiling MEASURED on an MI355X of this pool next to the vendor-stated peak — hipBLASLt (torch.matmul) on the step's
                        # bf16 shapes with random operands: 1202-1392 TFLOP/s (zero-filled operands: up to 2034: the difference is the clock under the power cap)
                        "measured_vendor_gemm_ceiling": {"tflops": 1391.6, "frac_of_it": round(ach / 1391.6, 3),
                                                         "what": "hipBLASLt bf16 NT 36864 x 12288 x 3072, random operands (its best of the step's shapes)",
                                                         "source": "profiles/r04_hipblaslt_ceiling.log (tools/hipblaslt_ceiling.py)"}}
                if args.graph:
                    roof["measured_on"] = "one eager step after the timed region (the timed steps are hipGraph replays, which run no host-side event code)"
            kernels = {k: {"ms_per_step": round(v["ms"] / prof_steps, 2), "launches_per_step": v["launches"] // prof_steps,
                           "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 else None,
                           "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 0) if v["ms"] > 0 else None}
                       for k, v in prof.items() if v["launches"]}
        out = {
            "metric": f"training images/sec (whole node), {dict(flux='Flux.1-dev', sd3='SD3-Medium', sdxl='SDXL', sd15='SD 1.5', pixart='PixArt-Sigma')[args.model]} "
                      f"{'ControlNet branch' if (args.model == 'pixart' and not args.lora) else ('full fine-tune' + (' + EMA' if cfg.use_ema else '')) if args.full else f'LoRA r{args.rank}'} {args.res}^2 train step",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2), "ms_per_step_stats": step_stats, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (+ fp8 e5m2 x e4m3 trunk Linears)" if getattr(args, "fp8", False) else "bf16", "data": "synthetic",
            "config": {"workload": desc,
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}", "hip_graph": bool(args.graph),
                       "gradient_checkpointing": (f"interval={cfg.gradient_checkpointing_interval} stride={cfg.gradient_checkpointing_segment_stride}"
                                                  if cfg.gradient_checkpointing else False)},
            "step_model_tflops": round(step_flops / (ms_per_step * 1e-3) / 1e12, 1),
            "step_frac_of_bf16_mfma_peak": round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "loss": round(loss_val, 5),
            "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
            "roofline": roof,
            "kernels": kernels,
            "cpu_baseline": None,
        }
        if getattr(args, "fp8", False):
            out["note"] = ("coverage row, not a speed row: the fp8-native trunk is byte-exact to the reference's quantisers but 7 % SLOWER than bf16 end to end "
                           "(400.4 vs 374.0 ms, same box: profiles/r05_pixart_2k_fp8_ab.txt) — the per-call amax + quantise passes cost more than the forward third "
                           "of the trunk's GEMMs gains on the fp8 pipe")
        if world > 1 or comm_rep is not None:
            out["comm"] = comm_rep
        pub = published_row(args)
        if pub is not None and world == 1:
            ref_ips = B / pub[1]
            out["vs_baseline"] = round(value / ref_ips, 3)
            out["published"] = {"row": f"SD3 LoRA r128 1024^2 bs 3, bf16, checkpointing mode {pub[0]!r} (example sd3.peft-lora, optimizer adamw_bf16; this run: optimizer {args.optimizer!r}" + (" — bf16 adapter values and bf16 gradients under the fused AdamWBF16, as in the example)" if args.optimizer == "adamw_bf16" else " — fused fp32 AdamW over the fp32 adapter arena, NOT the example's optimizer)"), "sec_per_step": pub[1],
                                "images_per_s": round(ref_ips, 3), "hardware": "1x H100 (the reference's own sweep; BASELINE.md §1)",
                                "source": "documentation/experimental/SEGMENTED_CHECKPOINTING.md:795-805", "this_run_sec_per_step": round(ms_per_step / 1e3, 4)}
        if args.model == "flux" and args.full:
            # context only (vs_baseline stays null: a 32-GPU multi-node DeepSpeed figure incl. its inter-node exchange is not this configuration)
            out["published_context"] = {"row": "Flux.1-dev (12B) full-rank, 1024 px buckets, batch 8 per accelerator, 4 nodes x 8 H100 SXM5, DeepSpeed: 15 s/step (anecdotal)",
                                        "images_per_s_per_gpu": round(8 / 15.0, 3), "source": "documentation/DISTRIBUTED.md:291-298 (BASELINE.md section 1)",
                                        "this_run_images_per_s_per_gpu": round(value / world, 3)}
        if world == 1 and not args.no_cpu_baseline and args.model == "flux":
            del trainer, plugin, batches
            torch.cuda.empty_cache()
            _log("flux: timed region done, host-core leg")
            out["cpu_baseline"], out["parity_at_config"] = cpu_baseline(args, dev)
            if out["parity_at_config"] is not None and args.layers == 19 and args.single_layers == 38 and not args.full and time.time() - _T0 < 240.0:
                # the same check at the configuration's REAL depth (19 + 38 blocks, all 380 adapter gradients): the figures the GPU suite asserts
                # (tests/test_baseline_shapes_gpu.py::test_flux_full_depth_step_matches_oracle) — the 1 + 1-block form above is what the host-core leg timed
                try:
                    import gc
                    from tests import parity_at_config as PC
                    gc.collect(); torch.cuda.empty_cache()
                    _log("flux: parity at full depth")
                    t0_ = time.time()
                    out["parity_at_config"] = {"full_depth": PC.flux_lora_full_depth(dev, rank=int(args.rank)), "one_plus_one_blocks_of_the_host_core_leg": out["parity_at_config"]}
                    out["parity_at_config"]["full_depth"]["seconds"] = round(time.time() - t0_, 1)
                    gc.collect(); torch.cuda.empty_cache()
                except Exception as e:            # noqa: BLE001
                    out["parity_at_config"] = {"full_depth": {"error": f"{type(e).__name__}: {e}"[:300]}, "one_plus_one_blocks_of_the_host_core_leg": out["parity_at_config"]}
        elif world == 1 and not args.no_cpu_baseline and args.model in ("sd15", "sdxl"):
            del trainer, plugin, batches
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline_unet(args, sd15=args.model == "sd15", lora=not args.full)
            from tests import parity_at_config as PC       # the oracle as the checker of the HIP component at this configuration (never the thing timed)
            out["parity_at_config"] = PC.unet(args.model, args.res, dev, lora=not args.full, rank=int(args.rank))
        elif world == 1 and not args.no_cpu_baseline and args.model in ("sd3", "pixart"):
            del trainer, plugin, batches
            torch.cuda.empty_cache()
            from tests import parity_at_config as PC
            if args.model == "pixart" and args.lora:
                pass                                  # (the LoRA-on-the-trunk mode is a secondary measurement: its parity is the GPU suite's, tests/test_pixart_model_gpu.py)
            elif args.model == "pixart":
                out["parity_at_config"] = PC.pixart_controlnet(args.res, dev)
            elif args.full:
                out["parity_at_config"] = PC.sd3_full(args.res, dev)
        return out
    return None


def _guarded_main():
    """a rank that fails must take the job down, never leave its peers waiting in a collective: print what failed (incl. the library's own last error), then exit
    non-zero WITHOUT running finalisers (a process-group destructor can itself block on the peers); the launcher (torch.distributed.run) then stops the other ranks.
    A rank that stops making progress is caught by the collective timeout set at init_process_group (below the driver's own limit)."""
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:          # noqa: BLE001
        import traceback
        traceback.print_exc()
        last = ""
        try:
            from simpletuner_amd import lib as _lib
            last = _lib.load().st355_last_error().decode("utf-8", "replace")
        except Exception:               # noqa: BLE001
            pass
        print(f"[bench] rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} FAILED: {type(e).__name__}: {e}; st355_last_error: {last!r}",
              file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(1)


if __name__ == "__main__":
    _guarded_main()
