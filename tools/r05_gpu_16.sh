#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pixart_model_gpu.py -q -s -k "fp8" 2>&1 | grep -v "amdgpu.ids" | grep "pixart\|passed\|failed\|Error\|error\|assert" | cut -c1-400 | tee gpurun_out/r05_call16_tests.log
