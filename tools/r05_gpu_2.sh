#!/bin/bash
# round 5, GPU call 2: SDXL-LoRA outlier probe, the full-depth Flux step + 20-step loss curve at depth, default bench line as the round's starting point
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/sdxl_lora_outlier_probe.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r05_sdxl_lora_outlier_probe.log | tail -40
timeout 900 python -m pytest tests/test_baseline_shapes_gpu.py -q -s -k "full_depth or baseline-depth" 2>&1 | grep -v "amdgpu.ids" | grep "parity\|passed\|failed\|Error\|error\|assert" | cut -c1-400 | tee gpurun_out/r05_flux_full_depth.log
timeout 600 python bench.py > gpurun_out/r05_bench_start.json 2> gpurun_out/r05_bench_start.log; tail -c 1500 gpurun_out/r05_bench_start.json
