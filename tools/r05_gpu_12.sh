#!/bin/bash
# round 5, GPU call 12: per-GPU batch of the SDXL-LoRA workload against the tile quantisation of its 32^2-level GEMMs (M = 1024 B rows, N = 1280: 4 B x 5 tiles of 256 x 256 over
# 256 CUs — batch 16 = 320 tiles = 1.25 rounds; batch 12 = 240 = 0.94; batch 24 = 480 = 1.88), hipGraph replay, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp gpurun_out/r05_sdxl_lora_b16_shapes.txt gpurun_out/r05_sdxl_lora_b16_shapes.keep 2>/dev/null
for b in 12 16 24 32 48; do
  timeout 400 python bench.py --model sdxl --lora --rank 16 --batch $b --graph --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r05_sdxl_lora_b${b}_graph_line.json 2> gpurun_out/r05_sdxl_lora_b${b}_graph.log
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05_sdxl_lora_b${b}_graph_line.json").read().strip().splitlines()[-1])
    print("batch ${b}:", d["ms_per_step"], "ms/step", d["value"], d["unit"], "frac", d.get("step_frac_of_bf16_mfma_peak"), "peak GiB", d.get("peak_hbm_gib"), "gemm", d["kernels"]["gemm"]["ms_per_step"], d["kernels"]["gemm"]["tflops"])
except Exception as e:
    print("batch ${b}: no line", e); print(open("gpurun_out/r05_sdxl_lora_b${b}_graph.log").read()[-600:])
PY
done
