#!/usr/bin/env python3
"""Times the thin LoRA GEMMs of the Flux step (T = x A^T: [M,3072] x [128,3072]; U = dY (sB): [M,9216] x [128,9216]) through ops.gemm.
Run twice: default (split-K 128x128 schedule) and ST355_THIN_SPLITK=0 (256x128 / 256x256 schedules), same process settings otherwise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpletuner_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in ((36864, 128, 3072), (32768, 128, 3072), (4096, 128, 3072), (36864, 128, 9216), (36864, 64, 3072)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = ops.gemm(x, w)
    ref = (x[:256].float() @ w.float().t())
    err = ((out[:256].float() - ref).norm() / ref.norm()).item()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(x, w, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{M}x{N}x{K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {(M * K * 2 + M * N * 2) / us / 1e3:7.1f} GB/s  rel_err {err:.2e}  splitk={os.environ.get('ST355_THIN_SPLITK', '1')}")
