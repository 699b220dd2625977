#!/bin/bash
# round 5, GPU call 1: (a) the five cases gated behind ST355_GPU_NOT_YET_RUN, (b) GEMM tile-group width sweep: rate + FETCH_SIZE + package power / sclk sampled mid-run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export ST355_GPU_NOT_YET_RUN=1
timeout 400 python -m pytest tests/test_flux_lora_sets_gpu.py "tests/test_flux_model_gpu.py::test_flux_tokenwise_timesteps_match_oracle" \
  "tests/test_sd3_model_gpu.py::test_sd3_tokenwise_timesteps_match_oracle" -q -s 2>&1 | grep -v "amdgpu.ids" | tail -60 | cut -c1-300 | tee gpurun_out/r05_first_call.log
unset ST355_GPU_NOT_YET_RUN
set -o pipefail
WIDTHS="2 4 8 16 32"
out=gpurun_out/r05_gemm_tile_order.log; : > $out
export TMPDIR=/tmp ST355_GEMM_IMPL=pq
for shape in 36864,12288,3072 36864,3072,12288 36864,3072,3072; do
  for g in $WIDTHS; do
    # sustained run (~2 s) with power / sclk sampled in the middle
    it=$(python -c "m,n,k=map(int,'$shape'.split(','));print(int(4.0/(2.0*m*n*k/1.25e15)))")
    (LAB_SHAPE=$shape LAB_ITERS=$it timeout 60 tools/gemm_lab_g$g --child > /tmp/pp_$g.log 2>&1 &)
    sleep 2.6
    p=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Package Power\|Socket Power" | sed 's/.*: //' | tr '\n' ' ')
    while pgrep -x gemm_lab_g$g > /dev/null; do sleep 0.2; done
    r=$(grep -i "tflop" /tmp/pp_$g.log | tail -1)
    d=gpurun_out/r05_tile_order_pmc_g${g}_$(echo $shape | tr ',' 'x'); rm -rf $d
    (cd /tmp && LAB_SHAPE=$shape LAB_ITERS=4 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$d -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/gemm_lab_g$g --child > /dev/null 2>&1)
    f=$(python - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("$d/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("FETCH_SIZE/launch %.3f GB (x2 on gfx950: MI355X_MICROARCH.md)" % (2 * 1024 * sum(v) / max(1, len(v)) / 1e9) if v else "no counter rows")
PY
)
    rm -rf $d
    echo "group $g | $shape | $r | power/sclk: $p | $f" | tee -a $out
  done
done
