#!/bin/bash
# second GPU-box session of round 3: the default bench line with progress log, the new tests (SD3.5, LoRA rank > 64, no-copies attention backward at head_dim 64 / 96),
# the RCCL shared-device probe, the attention lab at SD3 / PixArt shapes, the reference's published SD3 rows.   usage: tools/r03_gpu_round2.sh <tag>
tag=${1:-r03c}
mkdir -p gpurun_out
timeout 700 python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
grep "^\[bench" gpurun_out/${tag}_bench.err | tail -40
tail -c 300 gpurun_out/${tag}_bench_line.json; echo
timeout 900 python -m pytest tests/test_sd3_model_gpu.py tests/test_ref_models_gpu.py tests/test_kernels_gpu.py tests/test_unet_kernels_gpu.py tests/test_pixart_model_gpu.py tests/test_unet_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "sd3 or sd35 or attention or attn or pixart or unet or reference" > gpurun_out/${tag}_pytest_new.log 2>&1
tail -6 gpurun_out/${tag}_pytest_new.log
timeout 150 python tools/probes/rccl_shared_gpu.py > gpurun_out/${tag}_rccl_shared_gpu.log 2>&1; echo "rccl probe rc=$?"; grep "rccl-shared-gpu" gpurun_out/${tag}_rccl_shared_gpu.log | head -4
for shp in "4 24 4327 64" "1 16 16384 96"; do
  timeout 200 tools/attn_lab $shp > "gpurun_out/${tag}_attn_lab_$(echo $shp | tr ' ' '_').log" 2>&1
  grep -E "bwd|differ|MISMATCH" "gpurun_out/${tag}_attn_lab_$(echo $shp | tr ' ' '_').log" | head -12
done
for mode in "" "--gradient-checkpointing" "--gradient-checkpointing --ckpt-interval 2 --ckpt-stride 4"; do
  n=$(echo "none$mode" | tr -d ' -' | cut -c1-40)
  timeout 400 python bench.py --model sd3 --rank 128 --batch 3 --optimizer adamw_bf16 --no-cpu-baseline --steps 10 --warmup 3 $mode > gpurun_out/${tag}_sd3_r128_bs3_${n}.json 2> gpurun_out/${tag}_sd3_r128_bs3_${n}.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_sd3_r128_bs3_${n}.json").read().strip().splitlines()[-1])
    print("sd3 r128 bs3 [$mode]", d["ms_per_step"], "ms/step", d["value"], "img/s", d.get("published"), "vs", d.get("vs_baseline"))
except Exception as e:
    print("sd3 r128 bs3 [$mode] FAILED", e); print(open("gpurun_out/${tag}_sd3_r128_bs3_${n}.err").read()[-800:])
PY
done
