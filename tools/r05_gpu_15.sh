#!/bin/bash
# round 5, GPU call 15: LoRA over the fp8-native PixArt trunk (new), the other PixArt model tests, and the bench line of that configuration
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pixart_model_gpu.py -q -s 2>&1 | grep -v "amdgpu.ids" | grep "pixart\|passed\|failed\|Error\|error\|assert" | cut -c1-400 | tee gpurun_out/r05_call15_tests.log
for f in "" "--fp8"; do
  timeout 300 python bench.py --model pixart --lora --rank 128 --res 1024 --batch 3 --graph $f --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/r05_pixart_lora_r128_bs3_graph${f/--/_}_line.json 2> gpurun_out/r05_pixart_lora_r128_bs3_graph${f/--/_}.log
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05_pixart_lora_r128_bs3_graph${f/--/_}_line.json").read().strip().splitlines()[-1])
    print("pixart lora r128 bs3 graph ${f}:", d["ms_per_step"], "ms/step", d["value"], d["unit"], "loss", d.get("loss"), {k: v["ms_per_step"] for k, v in d["kernels"].items()})
except Exception as e:
    print("no line ${f}", e); print(open("gpurun_out/r05_pixart_lora_r128_bs3_graph${f/--/_}.log").read()[-800:])
PY
done
