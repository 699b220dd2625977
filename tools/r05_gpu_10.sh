#!/bin/bash
# round 5, GPU call 10: split-K slice count of the weight-gradient (TN) GEMMs — the old "about 384 workgroups" rule (ST355_TN_KS=-1) against the cost model (0) and forced
# counts, in the lab on the SD3-Medium / SDXL / Flux full-rank shapes, then in the SD3 full fine-tune step and the SDXL full fine-tune step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r05_tn_slice_count.log; : > $L
for shape in 6144,1536,32768 1536,6144,32768 4608,1536,32768 1536,1536,32768 3072,3072,36864 1280,1280,16384 640,640,65536 1280,5120,16384; do
  for ks in -1 0 3 5 7 9; do
    ST355_TN_KS=$ks LAB_TN=$shape timeout 60 tools/gemm_lab --child 2>&1 | grep "TN" >> $L
  done
done
cat $L | cut -c1-150
run() { tag=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline > gpurun_out/r05_${tag}_line.json 2> gpurun_out/r05_${tag}.log; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05_${tag}_line.json").read().strip().splitlines()[-1])
    print("${tag}:", d["ms_per_step"], "ms/step", d["value"], d["unit"], "loss", d.get("loss"), "gemm", d["kernels"]["gemm"])
except Exception as e:
    print("${tag}: no line", e); print(open("gpurun_out/r05_${tag}.log").read()[-1200:])
PY
}
for m in -1 0 -1 0; do ST355_TN_KS=$m run sd3_full_b8_tnks$m --model sd3 --full --batch 8 --steps 4 --warmup 2; done
for m in -1 0; do ST355_TN_KS=$m run sdxl_full_b4_graph_tnks$m --model sdxl --batch 4 --graph --steps 4 --warmup 2; done
