#!/bin/bash
# round 5, GPU call 11: per-shape launch table of the SDXL-LoRA step (r16, 1024^2, per-GPU batch 16: the metric's SDXL half) — where its GEMM class (246 ms at 815 TFLOP/s) goes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 400 python bench.py --model sdxl --lora --rank 16 --batch 16 --steps 3 --warmup 2 --no-cpu-baseline --prof-dump gpurun_out/r05_sdxl_lora_b16_dump.csv > gpurun_out/r05_sdxl_lora_b16_eager_line.json 2> gpurun_out/r05_sdxl_lora_b16_eager.log
python tools/prof_shapes.py gpurun_out/r05_sdxl_lora_b16_dump.csv 3 > gpurun_out/r05_sdxl_lora_b16_shapes.txt; head -75 gpurun_out/r05_sdxl_lora_b16_shapes.txt | cut -c1-140
rm -f gpurun_out/r05_sdxl_lora_b16_dump.csv
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_sdxl_lora_b16_eager_line.json").read().strip().splitlines()[-1])
print("eager:", d["ms_per_step"], "ms/step", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
