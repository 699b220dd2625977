#!/usr/bin/env python3
"""Times the rank-space projections (N = 64 / 128) of the Flux / SDXL-LoRA steps through ops.gemm and checks them against fp32: run under ST355_GEMM_ROWS=0 / 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simpletuner_amd import ops
dev = torch.device("cuda:0")
shapes = ((32768, 64, 1280), (16384, 64, 1280), (131072, 64, 640), (32768, 128, 3840), (32768, 128, 1280), (131072, 128, 1920), (36864, 128, 3072), (36864, 128, 9216), (36864, 64, 3072), (16421, 64, 1280), (5000, 128, 320))
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = ops.gemm(x, w)
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.arange(M - 300, M, device=dev)])
    ref = x[rows].float() @ w.float().t()
    err = ((out[rows].float() - ref).norm() / ref.norm()).item()
    worst = (out[rows].float() - ref).abs().max().item()
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(10):
        flush.zero_()                                   # operands leave the Infinity Cache between launches, as in the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(x, w, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts)[len(ts) // 2]
    print(f"{M}x{N}x{K}: {us:8.1f} us  {(M * K * 2 + M * N * 2) / us / 1e3:7.1f} GB/s  rel_err {err:.2e} max_abs {worst:.3f}  ROWS={os.environ.get('ST355_GEMM_ROWS', '0')}", flush=True)
    del x, w, out, flush
