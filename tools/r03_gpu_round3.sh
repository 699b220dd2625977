#!/bin/bash
# third GPU-box session of round 3: Flux TREAD test, the reference's published SD3 LoRA r128 bs 3 rows, secondary workloads re-measured in round-3 code
tag=${1:-r03d}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_flux_model_gpu.py -m gpu -q -x -s -p no:cacheprovider -k "tread or checkpointed or step_matches" > gpurun_out/${tag}_pytest_flux_tread.log 2>&1
grep -E "tread flux|passed|failed|Error" gpurun_out/${tag}_pytest_flux_tread.log | tail -14
run() {  # name, args...
  n=$1; shift
  timeout 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_${n}.json 2> gpurun_out/${tag}_${n}.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_${n}.json").read().strip().splitlines()[-1])
    k = sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:7]
    print("${n}:", d["ms_per_step"], "ms/step", d["value"], "img/s frac", d["step_frac_of_bf16_mfma_peak"], "gemm TF", d["roofline"]["achieved"] if d.get("roofline") else None, d.get("published"), "vs_baseline", d.get("vs_baseline"))
    print("   ", [(a, b["ms_per_step"], b.get("tflops")) for a, b in k])
except Exception as e:
    print("${n} FAILED", e); print(open("gpurun_out/${tag}_${n}.err").read()[-600:])
PY
}
run sd3_r128_bs3_none --model sd3 --rank 128 --batch 3 --steps 10 --warmup 3
run sd3_r128_bs3_layer --model sd3 --rank 128 --batch 3 --steps 10 --warmup 3 --gradient-checkpointing
run sd3_r128_bs3_seg2_stride4 --model sd3 --rank 128 --batch 3 --steps 10 --warmup 3 --gradient-checkpointing --ckpt-interval 2 --ckpt-stride 4
run sd3_full_b8 --model sd3 --full --steps 5 --warmup 2
run sdxl_lora_b4_graph --model sdxl --lora --rank 16 --batch 4 --graph --steps 10 --warmup 3
run sdxl_full_b4_graph --model sdxl --batch 4 --graph --steps 10 --warmup 3
run pixart_2k --model pixart --res 2048 --batch 1 --steps 5 --warmup 2
