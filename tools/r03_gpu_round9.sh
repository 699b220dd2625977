#!/bin/bash
# ninth GPU-box session of round 3: the part of the GPU suite that follows the Flux TREAD tests (the r03p sequence stopped there with -x), after the fix
tag=${1:-r03q}
mkdir -p gpurun_out
timeout 285 python -m pytest tests/test_flux_model_gpu.py tests/test_fp8_linear.py tests/test_golden_gpu.py tests/test_kernels_gpu.py tests/test_loss_golden.py tests/test_optimizer_state_gpu.py \
  tests/test_parity_at_config_gpu.py tests/test_pixart_model_gpu.py tests/test_prepare_batch_gpu.py tests/test_ref_models_gpu.py tests/test_sd3_model_gpu.py tests/test_trainer_graph_gpu.py \
  tests/test_unet_kernels_gpu.py tests/test_unet_model_gpu.py tests/test_vae_gpu.py tests/test_xm_gpu.py -m gpu -q -p no:cacheprovider \
  --deselect tests/test_flux_model_gpu.py::test_flux_step_matches_oracle --deselect tests/test_flux_model_gpu.py::test_flux_loss_curve_matches_oracle_adamw \
  --deselect tests/test_flux_model_gpu.py::test_ema_and_clipping_in_the_loop --deselect tests/test_flux_model_gpu.py::test_checkpointed_gradients_equal_direct_gradients \
  --deselect tests/test_flux_model_gpu.py::test_flux_attention_masked_training_matches_oracle > gpurun_out/${tag}_pytest_rest.log 2>&1
grep -E "passed|failed" gpurun_out/${tag}_pytest_rest.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/${tag}_pytest_rest.log | head -10
