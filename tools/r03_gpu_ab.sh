#!/bin/bash
# env-switch A/B runs on one box: the persistent GEMM schedule and the 256x256 tile threshold on the secondary workloads
tag=${1:-r03i}
mkdir -p gpurun_out
run() {  # label, env, args...
  lbl=$1; shift; envs=$1; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_$lbl.json 2> gpurun_out/${tag}_$lbl.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_$lbl.json").read().strip().splitlines()[-1])
    print("$lbl [$envs]:", d["ms_per_step"], "ms/step", d["value"], "img/s  gemm", d["kernels"].get("gemm"))
except Exception as e:
    print("$lbl FAILED", e); print(open("gpurun_out/${tag}_$lbl.err").read()[-400:])
PY
}
for pz in 0 1; do
  run sd3full_pz$pz "ST355_GEMM_PERSIST=$pz" --model sd3 --full --steps 5 --warmup 2
  run sdxl16_pz$pz "ST355_GEMM_PERSIST=$pz" --model sdxl --lora --rank 16 --batch 16 --graph --steps 5 --warmup 2
done
for mt in 200 128 64; do
  run sdxl4_mt$mt "ST355_GEMM_MIN_TILES=$mt" --model sdxl --lora --rank 16 --batch 4 --graph --steps 8 --warmup 3
  run sd3full_mt$mt "ST355_GEMM_MIN_TILES=$mt" --model sd3 --full --steps 5 --warmup 2
done
