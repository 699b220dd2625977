#!/usr/bin/env python3
"""Wave-quantisation model over a `bench.py --prof-dump` CSV: for every GEMM shape of the step, tiles of 256x256 -> waves of 256 CUs ->
quantisation efficiency, measured TFLOP/s and TFLOP/s normalised by that efficiency (= the rate while CUs are occupied).  Separates the two
in-step losses of the GEMM class: partial last waves (fixable by stream-K / larger M) vs per-tile overhead (prologue + epilogue not overlapped
with the K loop; visible as a low normalised rate on short-K, heavy-epilogue shapes).

    python tools/gemm_wave_model.py gpurun_out/flux_shapes.csv [n_cus=256] [tile=256]
"""
import csv
import math
import re
import sys
from collections import defaultdict

path = sys.argv[1]
cus = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 256
agg = defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(path)):
    if int(r["class"]) != 0:
        continue
    a = agg[r["tag"]]
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["flops"])
rows = []
for tag, (n, ms, fl) in agg.items():
    m = re.match(r"(\d+)(?:&(\d+))?x(\d+)x(\d+)\+(\d+) e(\d)", tag)
    if not m:
        continue
    m1, m2, nn, k, k2, epi = m.groups()
    row_tiles = math.ceil(int(m1) / tile) + (math.ceil(int(m2) / tile) if m2 else 0)
    tiles = row_tiles * math.ceil(int(nn) / tile)
    waves = tiles / cus
    q = waves / math.ceil(waves)
    tf = fl / ms / 1e9
    rows.append((ms, tag, n, tiles, waves, q, tf, tf / q, (int(k) + int(k2)) // 64))
tot = sum(r[0] for r in rows)
print(f"{'shape (MxNxK+K2 eEPI)':36s} {'ms':>8s} {'%':>5s} {'tiles':>6s} {'waves':>6s} {'q-eff':>6s} {'k-tiles':>7s} {'TF/s':>7s} {'TF/s/q':>7s}")
for ms, tag, n, tiles, waves, q, tf, tfq, kt in sorted(rows, reverse=True):
    print(f"{tag:36s} {ms:8.1f} {100 * ms / tot:5.1f} {tiles:6d} {waves:6.2f} {q:6.3f} {kt:7d} {tf:7.0f} {tfq:7.0f}")
print(f"GEMM total {tot:.1f} ms; time-weighted quantisation efficiency {sum(r[0] * r[5] for r in rows) / tot:.3f}; "
      f"time-weighted TF/s {sum(r[0] * r[6] for r in rows) / tot:.0f}, normalised {sum(r[0] * r[7] for r in rows) / tot:.0f}")
