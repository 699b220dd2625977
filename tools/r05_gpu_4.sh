#!/bin/bash
# round 5, GPU call 4: (a) head_dim-96 bodies with 5 contraction k-steps (valid head_dim <= 80) vs the committed 6-k-step bodies at PixArt-Sigma's 2K shape,
# (b) the output-residual attention backward: kernel test, the SDXL-LoRA probe again, parity at configs[0] / [1], UNet suites, graph capture,
# (c) SD3: published row under adamw_bf16, mixed buckets under per-bucket graph replay, the ungroup rule A/B on the batch-8 full fine-tune
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r05_attn_lab_hd96_contraction_k_steps.log; : > $L
for v in h96base h96a h96b h96c; do
  echo "=== $v (LAB_DVALID=72, B1 H16 S16384 d96)" >> $L
  LAB_DVALID=72 LAB_ITERS=6 timeout 120 tools/attn_lab_$v 1 16 16384 96 2>&1 | grep -v "generation 1\|row-major" >> $L
done
echo "=== h96a at SD 1.5's padded 80 (LAB_DVALID=80, B2 H8 S4096 d96)" >> $L
LAB_DVALID=80 LAB_ITERS=4 timeout 120 tools/attn_lab_h96a 2 8 4096 96 2>&1 | grep -i "mismatch\|identical\|rounding\|ok" >> $L
grep -i "TFLOP\|===\|MISMATCH\|identical\|rounding" $L | cut -c1-160
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_kernels_gpu.py tests/test_unet_model_gpu.py tests/test_trainer_graph_gpu.py "tests/test_parity_at_config_gpu.py::test_sd15_lora_r16_512_true_architecture" "tests/test_parity_at_config_gpu.py::test_sdxl_1024_true_architecture" -q -x -s 2>&1 | grep -v "amdgpu.ids" | grep "common component\|parity@config\|passed\|failed\|Error\|error\|assert" | cut -c1-900 | tee gpurun_out/r05_call4_tests.log
timeout 300 python tools/sdxl_lora_outlier_probe.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_sdxl_lora_outlier_probe_after.log; grep "==\|stage A \|lora_A" gpurun_out/r05_sdxl_lora_outlier_probe_after.log | cut -c1-250
run() { tag=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline > gpurun_out/r05_${tag}_line.json 2> gpurun_out/r05_${tag}.log; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05_${tag}_line.json").read().strip().splitlines()[-1])
    print("${tag}:", d["ms_per_step"], "ms/step", d["value"], d["unit"], "frac", d.get("step_frac_of_bf16_mfma_peak"), "vs_baseline", d.get("vs_baseline"), "gemm", d["kernels"]["gemm"])
except Exception as e:
    print("${tag}: no line", e); print(open("gpurun_out/r05_${tag}.log").read()[-1500:])
PY
}
run sd3_r128_bs3_graph_adamw_bf16 --model sd3 --rank 128 --batch 3 --graph --optimizer adamw_bf16 --steps 8 --warmup 3
run sd3_r128_bs3_graph_fp32_adamw --model sd3 --rank 128 --batch 3 --graph --steps 8 --warmup 3
run sd3_full_buckets_graph --model sd3 --full --batch 8 --buckets --graph --steps 5 --warmup 2
run sd3_full_b8_ungroup --model sd3 --full --batch 8 --steps 4 --warmup 2
ST355_GEMM_UNGROUP=0 run sd3_full_b8_grouped --model sd3 --full --batch 8 --steps 4 --warmup 2
