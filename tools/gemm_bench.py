#!/usr/bin/env python3
"""GEMM schedule microbench (random bf16 operands, as the guide demands): python tools/gemm_bench.py  -> TFLOP/s per shape.
Select the schedule with ST355_GEMM_IMPL=s2|p3|(unset: all)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simpletuner_amd import ops  # noqa: E402

SHAPES = [(4608, 3072, 3072), (4608, 9216, 3072), (4608, 12288, 3072), (4608, 3072, 12288), (4096, 3072, 3072), (512, 3072, 3072),
          (18432, 3072, 3072), (18432, 12288, 3072), (8192, 8192, 8192), (4608, 128, 3072), (4608, 128, 9216)]


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print("impl =", os.environ.get("ST355_GEMM_IMPL", "default"))
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(a, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for _ in range(iters):
            ops.gemm(a, w, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"  {M:6d} x {N:6d} x {K:6d}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
