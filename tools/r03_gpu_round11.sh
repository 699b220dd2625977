#!/bin/bash
# eleventh GPU-box session of round 3: the two-level reduction of the q / k RMSNorm weight gradient (full-rank Flux, SD3.5 full fine-tune): parity + the step time it buys
tag=${1:-r03s}
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_flux_full_rank_gpu.py tests/test_sd3_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "full" > gpurun_out/${tag}_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/${tag}_pytest.log | tail -2
timeout 100 python bench.py --model flux --full --batch 8 --steps 3 --warmup 1 --optimizer adamw_bf16 --gradient-checkpointing --ckpt-interval 3 --ckpt-stride 4 --no-cpu-baseline --no-secondary \
  > gpurun_out/${tag}_flux_full_rank_i3_s4.json 2> gpurun_out/${tag}_flux_full_rank.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_flux_full_rank_i3_s4.json").read().strip().splitlines()[-1])
    print("interval 3 stride 4:", d["value"], d["ms_per_step"], "peak GiB", d["peak_hbm_gib"], "loss", d["loss"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/${tag}_flux_full_rank.err").read()[-600:])
PY
