#!/bin/bash
# round 5, GPU call 5: kernel / UNet / PixArt / graph / parity-at-config suites on the changed attention paths (output residual; head_dim-96 bodies with 5 contraction
# k-steps), then SD3 published row under adamw_bf16, mixed buckets under per-bucket graph replay, PixArt-Sigma 2K, and the Flux headline (tile group 4 for K >= 8192)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_unet_kernels_gpu.py tests/test_unet_model_gpu.py tests/test_trainer_graph_gpu.py tests/test_pixart_model_gpu.py tests/test_parity_at_config_gpu.py "tests/test_baseline_shapes_gpu.py::test_attention_at_baseline_shapes" -q -s 2>&1 | grep -v "amdgpu.ids" | grep "common component\|O vs exact\|parity@config\|passed\|failed\|Error\|error\|assert\|FAILED" | cut -c1-1200 | tee gpurun_out/r05_call5_tests.log
run() { tag=$1; shift; timeout 500 python bench.py "$@" --no-cpu-baseline > gpurun_out/r05_${tag}_line.json 2> gpurun_out/r05_${tag}.log; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05_${tag}_line.json").read().strip().splitlines()[-1])
    print("${tag}:", d["ms_per_step"], "ms/step", d["value"], d["unit"], "frac", d.get("step_frac_of_bf16_mfma_peak"), "vs_baseline", d.get("vs_baseline"), "peak GiB", d.get("peak_hbm_gib"))
    print("    ", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
except Exception as e:
    print("${tag}: no line", e); print(open("gpurun_out/r05_${tag}.log").read()[-1500:])
PY
}
run sd3_r128_bs3_graph_adamw_bf16 --model sd3 --rank 128 --batch 3 --graph --optimizer adamw_bf16 --steps 8 --warmup 3
run sd3_full_buckets_graph --model sd3 --full --batch 8 --buckets --graph --steps 5 --warmup 2
run sd3_full_buckets_eager --model sd3 --full --batch 8 --buckets --steps 5 --warmup 2
run pixart_2k --model pixart --res 2048 --steps 4 --warmup 2
run flux_headline_quick --steps 6 --warmup 2 --no-secondary
