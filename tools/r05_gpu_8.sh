#!/bin/bash
# round 5, GPU call 8: the Flux step with the single block's two-segment proj_out (K + K2 = 15360) on the one-tile-per-workgroup schedule (new) vs the persistent one (old), same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mode in 1 0 1 0; do
  ST355_GEMM_PZ_LONGK=$mode timeout 300 python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --prof-dump gpurun_out/r05_flux_longk${mode}_dump.csv > gpurun_out/r05_flux_longk${mode}_line.json 2> gpurun_out/r05_flux_longk${mode}.log
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_flux_longk${mode}_line.json").read().strip().splitlines()[-1])
print("ST355_GEMM_PZ_LONGK=${mode}:", d["ms_per_step"], "ms/step", d["value"], "images/s; gemm", d["kernels"]["gemm"])
PY
  python tools/prof_shapes.py gpurun_out/r05_flux_longk${mode}_dump.csv 5 | grep "3072x3072+12288\|12288x3072+0 e1 \|3072x12288+0 e0" | head -4
  rm -f gpurun_out/r05_flux_longk${mode}_dump.csv
done
