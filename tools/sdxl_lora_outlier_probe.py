"""Lab probe (not product, not a test): which kernel owns the SDXL-LoRA true-depth outlier?

tests/parity_at_config.unet("sdxl", 1024, lora=True) reports its worst adapter gradient at up_blocks.0.attentions.2.transformer_blocks.{3,7}.attn1.{to_q,to_k}.lora_A
(rel-L2 0.11 against the fp32 oracle; 3.1 x what torch's bf16 autograd of the same restatement sits at).  That tensor is behind the head_dim-64 attention backward
and the rank-space products (U = dy sB, dA = U^T x).  This probe runs the same network once, captures the OPERANDS of that layer's backward on the HIP path (x, dy of
the fused to_q/to_k/to_v projection; q/k/v, O, dO of the attention) and recomputes each stage from the captured operands in fp64 on the device:

  stage R  rank-space products alone : dA_hip vs (dy_hip sB)^T x_hip in fp64               -> error owned by k_skinny_tn_mfma + the U GEMM
  stage A  attention backward alone  : dq/dk/dv_hip vs fp64 softmax-attention backward from the captured q, k, v, dO
           A' the same fp64 backward but with delta = rowsum(dO * O_bf16) (the rounded O the flash backward reads) -> how much of A is the delta inconsistency
  stage P  propagated               : dA from (dy_fp64 of stage A) vs dA from dy_hip, both fp64 products      -> what the attention backward's error does to THIS tensor
  total                              : dA_hip vs the fp32 oracle of the whole network (the number parity_at_config reports)

usage (GPU box): python tools/sdxl_lora_outlier_probe.py  > gpurun_out/r05_sdxl_lora_outlier_probe.log
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

BF16 = torch.bfloat16


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def attn_bwd_ref(q, k, v, dO, heads, O_for_delta=None):
    """fp64 softmax attention backward; q/k/v/dO [S, heads*hd].  O_for_delta: use this (rounded) O in delta = rowsum(dO * O) instead of the exact one."""
    S, C = q.shape
    hd = C // heads
    sp = lambda t: t.double().view(S, heads, hd).transpose(0, 1)         # [h, S, hd]
    Q, K, V, dOh = sp(q), sp(k), sp(v), sp(dO)
    scale = hd ** -0.5
    P = torch.softmax(Q @ K.transpose(1, 2) * scale, dim=-1)
    O = P @ V
    dV = P.transpose(1, 2) @ dOh
    dP = dOh @ V.transpose(1, 2)
    Od = O if O_for_delta is None else sp(O_for_delta)
    delta = (dOh * Od).sum(-1, keepdim=True)
    dS = P * (dP - delta)
    dQ = dS @ K * scale
    dK = dS.transpose(1, 2) @ Q * scale
    mg = lambda t: t.transpose(0, 1).reshape(S, C)
    return mg(dQ), mg(dK), mg(dV), mg(O)


def main():
    from oracle.unet import UNetConfig, unet_forward
    from simpletuner_amd.unet.unet import UNet2DConditionModel

    dev = torch.device("cuda:0")
    seed, rank, res = 4, 16, 1024
    ocfg = UNetConfig()
    m = UNet2DConditionModel(device=dev)
    m.init_synthetic(seed)
    lat = res // 8
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 4, lat, lat, generator=g).to(BF16)
    target = torch.randn(1, 4, lat, lat, generator=g)
    ctx = torch.randn(1, 77, ocfg.cross_attention_dim, generator=g).to(BF16)
    t = torch.tensor([417.0])
    te = torch.randn(1, 1280, generator=g).to(BF16)
    ti = torch.tensor([[float(res), float(res), 0.0, 0.0, float(res), float(res)]]).to(BF16)
    alpha = float(rank)
    m.add_lora_adapter(rank=rank, alpha=alpha, seed=seed + 1, init_b_std=0.02)
    watch = [f"up_blocks.0.attentions.2.transformer_blocks.{k}.attn1." for k in (3, 7)] + ["down_blocks.1.attentions.0.transformer_blocks.0.attn1.",
                                                                                           "mid_block.attentions.0.transformer_blocks.5.attn1."]
    cap = {}
    pending_attn = []

    def probe(kind, name, d):
        if kind == "attn":
            pending_attn.append(d)          # the attention backward runs just before the projection that produced its q / k / v
        elif kind == "linear":
            for w in watch:
                if name.startswith(w + "to_q"):
                    a = [p for p in pending_attn if p["self_attn"] and p["qsrc"].data_ptr() == d["y"].data_ptr()]
                    cap[w] = dict(x=d["x"].clone(), dy=d["dy"].clone(), attn={k: (v.clone() if torch.is_tensor(v) else v) for k, v in a[-1].items()} if a else None)
            pending_attn.clear()

    m._probe = probe
    out = m(x.to(dev), t.to(dev), ctx.to(dev), None, added_cond_kwargs={"text_embeds": te.to(dev), "time_ids": ti.to(dev)}, return_dict=False)[0]
    ((out.float() - target.to(dev)) ** 2).mean().backward()
    torch.cuda.synchronize()
    m._probe = None
    # whole-network fp32 oracle (autograd) for the total
    P = {k: v.float().to(dev) for k, v in m.diffusers_state_dict().items()}
    lp = {n: p.detach().float().clone().requires_grad_(True) for n, p in m.named_parameters() if ".lora_" in n}
    Pe = dict(P)
    for n in lp:
        if ".lora_A." in n:
            base = n.replace(".lora_A.default.weight", "")
            Pe[base + ".weight"] = P[base + ".weight"] + (alpha / rank) * lp[base + ".lora_B.default.weight"] @ lp[n]
    ref = unet_forward(Pe, ocfg, x.float().to(dev), t.to(dev), ctx.float().to(dev), {"text_embeds": te.float().to(dev), "time_ids": ti.float().to(dev)})
    ((ref - target.to(dev)) ** 2).mean().backward()
    hip = dict(m.named_parameters())
    print(f"prediction rel-L2 vs fp32 oracle: {rel(out, ref.detach()):.3e}")
    for w in watch:
        c = cap.get(w)
        if c is None:
            print(f"{w}: not captured"); continue
        xh, dy, at = c["x"], c["dy"], c["attn"]
        C = xh.shape[1]
        heads = at["heads"]
        print(f"== {w}  tokens {xh.shape[0]}  C {C}  heads {heads}  hd {C // heads}")
        q, k, v = (at["qsrc"][:, i * C:(i + 1) * C] for i in range(3))
        dq_h, dk_h, dv_h = (at["dq_src"][:, i * C:(i + 1) * C] for i in range(3))
        dQ, dK, dV, Oex = attn_bwd_ref(q, k, v, at["dO"], heads)
        dQr, dKr, dVr, _ = attn_bwd_ref(q, k, v, at["dO"], heads, O_for_delta=at["O"])
        print(f"  stage A  attention backward from captured q,k,v,dO   : dQ {rel(dq_h, dQ):.3e}  dK {rel(dk_h, dK):.3e}  dV {rel(dv_h, dV):.3e}   (O_hip vs exact {rel(at['O'], Oex):.3e})")
        print(f"  stage A' fp64 backward with delta from the bf16 O     : dQ {rel(dQr, dQ):.3e}  dK {rel(dKr, dK):.3e}   | hip vs A': dQ {rel(dq_h, dQr):.3e}  dK {rel(dk_h, dKr):.3e}")
        print(f"           column sums over keys, |sum_j dK_j| / sum_j |dK_j| : exact {dK.sum(0).norm().item() / dK.norm(dim=1).sum().item():.3e}  "
              f"hip {dk_h.double().sum(0).norm().item() / dk_h.double().norm(dim=1).sum().item():.3e}  A' {dKr.sum(0).norm().item() / dKr.norm(dim=1).sum().item():.3e}")
        xm = xh.double().mean(0)
        print(f"           common component of x over tokens: |mean_j x_j| / rms_j |x_j| = {xm.norm().item() / xh.double().norm(dim=1).pow(2).mean().sqrt().item():.3f}")
        dy_ref = torch.cat([dQ, dK, dV], dim=1)
        for j, nm in enumerate(("to_q", "to_k", "to_v")):
            A = hip[w + nm + ".lora_A.default.weight"]; B = hip[w + nm + ".lora_B.default.weight"]
            sB = (alpha / rank) * B.detach().double()
            dA_hip = A.grad.double()
            dyj = dy[:, j * C:(j + 1) * C].double()
            dA_loc = (dyj @ sB).t() @ xh.double()
            dA_from_ref_dy = (dy_ref[:, j * C:(j + 1) * C] @ sB).t() @ xh.double()
            tot = rel(dA_hip, lp[w + nm + ".lora_A.default.weight"].grad)
            dB_hip = B.grad.double()
            dB_loc = (alpha / rank) * dyj.t() @ (xh.double() @ A.detach().double().t())
            print(f"  {nm}.lora_A: total vs oracle {tot:.3e} | stage R (rank-space products alone) {rel(dA_hip, dA_loc):.3e} | stage P (attention-backward error -> this tensor) "
                  f"{rel(dA_loc, dA_from_ref_dy):.3e}   lora_B: total {rel(dB_hip, lp[w + nm + '.lora_B.default.weight'].grad):.3e} stage R {rel(dB_hip, dB_loc):.3e}")


if __name__ == "__main__":
    main()
