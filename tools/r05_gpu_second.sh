#!/bin/bash
# second GPU call of the next round: the Flux step with the wider adapter sets (never timed: they were built after round 4's GPU budget was spent).
#   all+ffs: every block on the host-sequenced path (feed-forward / proj_mlp / proj_out adapters as K-extensions); tiny: the backward returns below single block 7
# usage: gpurun --timeout 900 -- tools/r05_gpu_second.sh     -> gpurun_out/r05_flux_<set>[_graph]_bench_line.json (copy what is kept into profiles/)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for t in all+ffs tiny; do
  for g in "" "--graph"; do
    tag=$(echo "${t}${g}" | tr '+' '_' | tr -d ' -')
    timeout 400 python bench.py --lora-target $t $g --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/r05_flux_${tag}_bench_line.json 2> gpurun_out/r05_flux_${tag}.log
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05_flux_${tag}_bench_line.json").read().strip().splitlines()[-1])
    print("${t} ${g}:", d["ms_per_step"], "ms/step,", d["value"], d["unit"], "gemm", d["kernels"]["gemm"]["tflops"], "TF")
except Exception as e:
    print("${t} ${g}: no line (", e, ")")
PY
  done
done
