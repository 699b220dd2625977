#!/bin/bash
# round 4: A/B of the round-quantisation GEMM tile choice (ST355_GEMM_ROUNDS) on the small-shape workloads, same box
cd /root/repo
O=gpurun_out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-secondary "$@" > $O/r04i_$name.json 2> $O/r04i_$name.err || echo "FAILED $name: $(tail -3 $O/r04i_$name.err)"; python - <<PY
import json
try:
    d = json.loads(open("$O/r04i_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["unit"], d["ms_per_step"], "ms", "loss", d["loss"], {k: (v["ms_per_step"], v["tflops"]) for k, v in d.get("kernels", {}).items() if k == "gemm"})
except Exception as e:
    print("$name: no line", e)
PY
}
for r in 0 1; do
  export ST355_GEMM_ROUNDS=$r
  run sdxl_lora_b16_rounds$r --model sdxl --lora --rank 16 --batch 16 --graph --steps 6 --warmup 3
  run sd3_full_b8_rounds$r --model sd3 --full --batch 8 --steps 6 --warmup 2
  run sdxl_full_b4_rounds$r --model sdxl --batch 4 --graph --steps 6 --warmup 3
done
