#!/bin/bash
# FIRST GPU CALL OF THE NEXT ROUND: the cases written after round 4's GPU budget was spent (emulator-checked on the CPU, never run on hardware):
#   * Flux adapter sets all+ffs+embedder (x_embedder: K = 64 projection with a K-extension) and ai-toolkit (modulation-Linear adapters: per-sample column sums, rank-space
#     gradients over M = batch rows) — tests/test_flux_lora_sets_gpu.py
#   * tokenwise timesteps at per-GPU batch 2 with rows that are no multiple of 256 — tests/test_{flux,sd3}_model_gpu.py
# usage: gpurun --timeout 300 -- tools/r05_gpu_first.sh     (then drop the ST355_GPU_NOT_YET_RUN gates of the cases that passed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export ST355_GPU_NOT_YET_RUN=1
timeout 280 python -m pytest tests/test_flux_lora_sets_gpu.py "tests/test_flux_model_gpu.py::test_flux_tokenwise_timesteps_match_oracle" \
  "tests/test_sd3_model_gpu.py::test_sd3_tokenwise_timesteps_match_oracle" -q -s 2>&1 | grep "parity\|passed\|failed\|Error\|error" | cut -c1-260 | tee gpurun_out/r05_first_call.log
