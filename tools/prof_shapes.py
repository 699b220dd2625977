#!/usr/bin/env python3
"""Summarise a bench.py --prof-dump CSV per (kernel class, shape tag): launches, total ms, TFLOP/s, GB/s.   python tools/prof_shapes.py dump.csv [steps]"""
import csv
import sys
from collections import defaultdict

CLASSES = ["gemm", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_prep", "ln_mod", "qk_rope", "skinny", "elementwise", "optim"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    a = agg[(int(r["class"]), r["tag"])]
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["flops"]); a[3] += float(r["bytes"])
tot = sum(a[1] for a in agg.values())
print(f"{'class':14s} {'shape':34s} {'n/step':>7s} {'ms/step':>9s} {'%':>6s} {'us/launch':>10s} {'TFLOP/s':>9s} {'GB/s':>8s}")
for (k, tag), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    ms = a[1]
    print(f"{CLASSES[k]:14s} {tag:34s} {a[0] / steps:7.0f} {ms / steps:9.3f} {100 * ms / tot:6.1f} {1e3 * ms / a[0]:10.1f} {a[2] / ms / 1e9:9.1f} {a[3] / ms / 1e6:8.0f}")
print(f"total {tot / steps:.2f} ms/step")
