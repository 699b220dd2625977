#!/bin/bash
# seventh GPU-box session of round 3: Flux full-rank training (gradient parity of every parameter, fused optimizers, checkpoint bit-equality), the
# modulation-scale gradient from LN(x) in the SD3 / PixArt full backward, and one full-size Flux.1-dev full-rank step time
tag=${1:-r03n}
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_flux_full_rank_gpu.py tests/test_sd3_model_gpu.py tests/test_pixart_model_gpu.py -m gpu -q -x -s -p no:cacheprovider \
  -k "full or controlnet" > gpurun_out/${tag}_pytest_full.log 2>&1
grep -E "parity\]|\[flux full|\[sd3 full|passed|failed|Error|error" gpurun_out/${tag}_pytest_full.log | tail -25
timeout 120 python -m pytest tests/test_flux_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "step_matches or block_c_entry" > gpurun_out/${tag}_pytest_lora.log 2>&1
tail -2 gpurun_out/${tag}_pytest_lora.log
timeout 400 python bench.py --model flux --full --batch 8 --steps 3 --warmup 1 --optimizer adamw_bf16 --gradient-checkpointing --no-cpu-baseline --no-secondary \
  > gpurun_out/${tag}_flux_full_rank_bench_line.json 2> gpurun_out/${tag}_flux_full_rank_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_flux_full_rank_bench_line.json").read().strip().splitlines()[-1])
    print("flux full-rank", d["value"], d["ms_per_step"], d["ms_per_step_stats"], "loss", d["loss"], d["step_model_tflops"], d["roofline"] and d["roofline"]["achieved"])
    print({k: v["ms_per_step"] for k, v in (d.get("kernels") or {}).items()})
except Exception as e:
    print("flux full-rank bench FAILED", e); print(open("gpurun_out/${tag}_flux_full_rank_bench.err").read()[-2500:])
PY
python -c "import torch; print('peak mem GB', 'n/a')"
