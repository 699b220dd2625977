#!/bin/bash
# one GPU-box session of round 3: the GPU suite (minus the at-shape file already green in this round), the default bench line, the rocprofv3 passes,
# the reference's published SD3 LoRA r128 bs3 rows.  usage (repo root, on the box): tools/r03_gpu_round.sh <tag>
tag=${1:-r03b}
R=$PWD
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_baseline_shapes_gpu.py -p no:cacheprovider > gpurun_out/${tag}_pytest_gpu.log 2>&1
tail -4 gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench_line.json; echo
for mode in "" "--gradient-checkpointing" "--gradient-checkpointing --ckpt-interval 2 --ckpt-stride 4"; do
  n=$(echo "none$mode" | tr -d ' -' | cut -c1-40)
  timeout 600 python bench.py --model sd3 --rank 128 --batch 3 --optimizer adamw_bf16 --no-cpu-baseline --steps 10 --warmup 3 $mode > gpurun_out/${tag}_sd3_r128_bs3_${n}.json 2> gpurun_out/${tag}_sd3_r128_bs3_${n}.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_sd3_r128_bs3_${n}.json").read().strip().splitlines()[-1])
    print("sd3 r128 bs3 [$mode]", d["ms_per_step"], "ms/step", d["value"], "img/s", d.get("published"), "vs", d.get("vs_baseline"))
except Exception as e:
    print("sd3 r128 bs3 [$mode] FAILED", e); print(open("gpurun_out/${tag}_sd3_r128_bs3_${n}.err").read()[-1500:])
PY
done
tools/profile_round.sh $tag > gpurun_out/${tag}_profile_round.log 2>&1
tail -3 gpurun_out/${tag}_profile_round.log
