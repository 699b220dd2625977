#!/bin/bash
# fifth GPU-box session of round 3: the Flux single-block C entry points (bit-equality with the host sequencing, parity at width / depth, step time A/B)
tag=${1:-r03f}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_flux_model_gpu.py tests/test_ref_models_gpu.py tests/test_golden_gpu.py tests/test_distributed_gpu.py tests/test_optimizer_state_gpu.py tests/test_prepare_batch_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/${tag}_pytest_blocks.log 2>&1
tail -4 gpurun_out/${tag}_pytest_blocks.log
timeout 600 python -m pytest tests/test_baseline_shapes_gpu.py -m gpu -q -x -s -p no:cacheprovider -k "flux_full_width or flux_full_depth" 2>&1 | grep -E "parity@config|passed|failed|Error" | tail -6
for abi in 0 1; do
  ST355_BLOCK_ABI=$abi timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > gpurun_out/${tag}_bench_abi$abi.json 2> gpurun_out/${tag}_bench_abi$abi.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_bench_abi$abi.json").read().strip().splitlines()[-1])
    print("ST355_BLOCK_ABI=$abi", d["ms_per_step"], d["ms_per_step_stats"]["median"], "loss", d["loss"])
except Exception as e:
    print("bench abi $abi FAILED", e); print(open("gpurun_out/${tag}_bench_abi$abi.err").read()[-800:])
PY
done
