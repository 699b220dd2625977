#!/usr/bin/env python3
"""Measured GEMM ceiling on this box ("confirm on the box with a hipBLASLt peak microbench and report that measured ceiling too"): torch.matmul
(ATen -> hipBLASLt / rocBLAS) and st355_gemm_bf16 on the same bf16 NT problems, with random and with zero-filled operands (zero-filled: no data toggling, the chip
is not power-limited — the gap between the two columns is what the power cap costs).  Measurement tool only: nothing in the product calls torch.matmul.

    python tools/hipblaslt_ceiling.py            # one MI355X
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simpletuner_amd import ops  # noqa: E402


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    print(f"{'M x N x K':>22} {'operands':>8} | torch.matmul (hipBLASLt)  |  st355_gemm_bf16")
    for (M, N, K) in ((8192, 8192, 8192), (8192, 8192, 12288), (36864, 12288, 3072), (36864, 3072, 12288), (36864, 3072, 3072)):
        for kind in ("random", "zero"):
            mk = (lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)) if kind == "random" else (lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16))
            a, w = mk(M, K), mk(N, K)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            t_t = timed(lambda: torch.matmul(a, w.t(), out=out))
            t_s = timed(lambda: ops.gemm(a, w, out=out))
            fl = 2.0 * M * N * K
            print(f"{M:>8} x {N:>5} x {K:>5} {kind:>8} | {t_t * 1e3:9.1f} us {fl / t_t / 1e9:8.1f} TFLOP/s | {t_s * 1e3:9.1f} us {fl / t_s / 1e9:8.1f} TFLOP/s", flush=True)
            del a, w, out


if __name__ == "__main__":
    main()
