#!/bin/bash
# round 5, GPU call 7: PixArt-Sigma 2K with the fp8-native trunk vs bf16 on the same box (configs[4]; last measured in r02); the SD3 published row under adamw_bf16 after
# the per-step blocking copy was removed; the Flux step with the wider adapter sets (never timed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 500 python bench.py "$@" --no-cpu-baseline > gpurun_out/r05_${tag}_line.json 2> gpurun_out/r05_${tag}.log; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05_${tag}_line.json").read().strip().splitlines()[-1])
    print("${tag}:", d["ms_per_step"], "ms/step", d["value"], d["unit"], "frac", d.get("step_frac_of_bf16_mfma_peak"), "vs_baseline", d.get("vs_baseline"), "loss", d.get("loss"))
    print("    ", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
except Exception as e:
    print("${tag}: no line", e); print(open("gpurun_out/r05_${tag}.log").read()[-1200:])
PY
}
run pixart_2k_bf16 --model pixart --res 2048 --steps 4 --warmup 2
run pixart_2k_fp8 --model pixart --res 2048 --fp8 --steps 4 --warmup 2
run sd3_r128_bs3_graph_adamw_bf16_b --model sd3 --rank 128 --batch 3 --graph --optimizer adamw_bf16 --steps 8 --warmup 3
run sd3_r128_bs3_graph_fp32_adamw_b --model sd3 --rank 128 --batch 3 --graph --steps 8 --warmup 3
run flux_all_ffs --lora-target all+ffs --steps 4 --warmup 2 --no-secondary
run flux_tiny --lora-target tiny --steps 4 --warmup 2 --no-secondary
