"""GroupNorm forward + backward at one SDXL shape, a few iterations — run under rocprofv3 --kernel-trace --stats to get the per-kernel durations:
   python tools/probes/gn_time.py B H W C [silu] [dadd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simpletuner_amd import ops
B, H, W, C = (int(a) for a in sys.argv[1:5])
d = torch.device("cuda:0")
n = ops.conv_grid_rows(B, H, W) if hasattr(ops, "conv_grid_rows") else B * (H + 2) * (W + 2) + 64
x = torch.zeros(n, C, device=d, dtype=torch.bfloat16)
x[:B * (H + 2) * (W + 2)].view(B, H + 2, W + 2, C)[:, 1:-1, 1:-1] = torch.randn(B, H, W, C, device=d).to(torch.bfloat16)
gamma = torch.ones(C, device=d, dtype=torch.bfloat16); beta = torch.zeros(C, device=d, dtype=torch.bfloat16)
dy = torch.randn(n, C, device=d).to(torch.bfloat16)
flush = torch.empty(1 << 30, dtype=torch.uint8, device=d)
for _ in range(4):
    flush.zero_()
    y, st = ops.groupnorm_fwd(x, gamma, beta, B, H, W, silu=True, out_tokens=False)
    flush.zero_()
    dx = ops.groupnorm_bwd(dy, x, gamma, beta, st, B, H, W, silu=True, dy_tokens=False)
torch.cuda.synchronize()
print("tensor MB", n * C * 2 / 1e6)
