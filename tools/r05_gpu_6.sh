#!/bin/bash
# round 5, GPU call 6: the suites touching the changed attention paths (call 5 named a test that does not exist: pytest ran nothing)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_unet_kernels_gpu.py tests/test_unet_model_gpu.py tests/test_trainer_graph_gpu.py tests/test_pixart_model_gpu.py tests/test_parity_at_config_gpu.py "tests/test_baseline_shapes_gpu.py::test_self_attention_at_baseline_shapes" "tests/test_baseline_shapes_gpu.py::test_pixart_cross_attention_at_2k" -q -s 2>&1 | grep -v "amdgpu.ids" | grep "common component\|O vs exact\|parity@config\|passed\|failed\|Error\|error\|assert\|FAILED" | cut -c1-1200 | tee gpurun_out/r05_call6_tests.log
