#!/bin/bash
# eighth GPU-box session of round 3: Flux.1-dev full-rank step under the segmented checkpoint plans (how much of the 288 GB can hold kept activations instead of recomputing them)
tag=${1:-r03o}
mkdir -p gpurun_out
for plan in "3 4" "2 3"; do
  set -- $plan
  timeout 300 python bench.py --model flux --full --batch 8 --steps 3 --warmup 1 --optimizer adamw_bf16 --gradient-checkpointing --ckpt-interval $1 --ckpt-stride $2 --no-cpu-baseline --no-secondary \
    > gpurun_out/${tag}_flux_full_rank_i$1_s$2.json 2> gpurun_out/${tag}_flux_full_rank_i$1_s$2.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_flux_full_rank_i$1_s$2.json").read().strip().splitlines()[-1])
    print("interval $1 stride $2:", d["value"], d["ms_per_step"], "peak GiB", d["peak_hbm_gib"], "loss", d["loss"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
except Exception as e:
    print("interval $1 stride $2 FAILED", e); print(open("gpurun_out/${tag}_flux_full_rank_i$1_s$2.err").read()[-600:])
PY
done
