#!/bin/bash
# round 4: the head_dim-64 workloads (SDXL-LoRA, SD3) with dq64<64> / dkv4<64>
cd /root/repo
O=gpurun_out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-secondary "$@" > $O/r04f_$name.json 2> $O/r04f_$name.err || echo "FAILED $name: $(tail -3 $O/r04f_$name.err)"; python - <<PY
import json
try:
    d = json.loads(open("$O/r04f_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["unit"], d["ms_per_step"], "ms", {k: (v["ms_per_step"], v["tflops"]) for k, v in d.get("kernels", {}).items() if k.startswith("attn") or k == "gemm"})
except Exception as e:
    print("$name: no line", e)
PY
}
run sdxl_lora_r16_b16_graph --model sdxl --lora --rank 16 --batch 16 --graph --steps 6 --warmup 3
run sd3_r128_bs3 --model sd3 --rank 128 --batch 3 --steps 8 --warmup 3
run sd3_r128_bs3_graph --model sd3 --rank 128 --batch 3 --steps 8 --warmup 3 --graph
run sd3_full_b8 --model sd3 --full --batch 8 --steps 6 --warmup 2
