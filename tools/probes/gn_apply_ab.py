"""A/B of the GroupNorm apply passes (ST355_GN_APPLY=1: flat-index kernels, default: row-walking kernels): run once per form in separate processes with
`python tools/probes/gn_apply_ab.py save <file>`, then `python tools/probes/gn_apply_ab.py cmp <a> <b>` — the outputs must be bit-identical."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch


def run():
    from simpletuner_amd import ops
    d = torch.device("cuda:0")
    out = {}
    g = torch.Generator(device=d).manual_seed(3)
    for (B, H, W, C, silu, tok) in [(2, 32, 32, 1280, True, False), (1, 64, 64, 640, True, False), (2, 16, 16, 2560, True, False), (3, 24, 40, 320, False, True),
                                    (1, 128, 128, 320, True, False), (2, 8, 8, 64, True, True), (1, 5, 7, 1920, True, False)]:
        n = ops.conv_grid_rows(B, H, W) if hasattr(ops, "conv_grid_rows") else B * (H + 2) * (W + 2) + 64
        x = torch.zeros(n, C, device=d, dtype=torch.bfloat16)
        xi = torch.randn(B, H, W, C, device=d, generator=g).to(torch.bfloat16)
        x[:B * (H + 2) * (W + 2)].view(B, H + 2, W + 2, C)[:, 1:-1, 1:-1] = xi
        gamma = (1 + 0.1 * torch.randn(C, device=d, generator=g)).to(torch.bfloat16); beta = (0.1 * torch.randn(C, device=d, generator=g)).to(torch.bfloat16)
        y, st = ops.groupnorm_fwd(x, gamma, beta, B, H, W, silu=silu, out_tokens=tok)
        if tok:
            dy = torch.randn(B * H * W, C, device=d, generator=g).to(torch.bfloat16)
        else:
            dy = torch.randn(n, C, device=d, generator=g).to(torch.bfloat16)          # border rows hold garbage on purpose: they must not reach dx
        dadd = torch.randn(n, C, device=d, generator=g).to(torch.bfloat16) if C == 640 else None
        dx = ops.groupnorm_bwd(dy, x, gamma, beta, st, B, H, W, silu=silu, dy_tokens=tok, dadd=dadd)
        torch.cuda.synchronize()
        key = f"{B}x{H}x{W}x{C}"
        out[key + "/y"] = y.cpu(); out[key + "/dx"] = (dx[0] if isinstance(dx, tuple) else dx).cpu()
    return out


if sys.argv[1] == "save":
    torch.save(run(), sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = [k for k in a if not torch.equal(a[k].view(torch.int16), b[k].view(torch.int16))]
    print("gn apply A/B:", "bit-identical on", len(a), "tensors" if not bad else f"MISMATCH {bad}")
    sys.exit(1 if bad else 0)
