"""Probe (GPU box, host cores only): fp32 torch linear + attention-shaped matmul throughput against the intra-op thread count — picks the thread count of
bench.py's cpu_baseline leg (256 logical cpus on the box; more threads is not always faster)."""
import os
import time

import torch

torch.manual_seed(0)
x = torch.randn(4608, 3072)
w = torch.randn(12288, 3072)
q = torch.randn(24, 4608, 128)
for n in (16, 32, 64, 128, os.cpu_count() or 1):
    torch.set_num_threads(n)
    torch.nn.functional.linear(x, w)                 # warm
    t0 = time.time()
    for _ in range(3):
        torch.nn.functional.linear(x, w)
    t_lin = (time.time() - t0) / 3
    t0 = time.time()
    s = torch.softmax(q @ q.transpose(1, 2), dim=-1) @ q
    t_att = time.time() - t0
    print(f"[cpu-threads] {n:4d} threads: linear 4608x3072x12288 {t_lin * 1e3:8.1f} ms = {2 * 4608 * 3072 * 12288 / t_lin / 1e12:6.2f} TFLOP/s; attention 24x4608x128 {t_att * 1e3:8.1f} ms", flush=True)
