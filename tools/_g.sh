cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["ms_per_step"], d["value"], d.get("loss"), {k: v["ms_per_step"] for k, v in d.get("kernels", {}).items()})
    for k, v in d.get("secondary", {}).items(): print("   ", k, v.get("ms_per_step"), v.get("value"), v.get("step_frac_of_bf16_mfma_peak"), v.get("error"))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for e in "X=1" "ST355_GEMM_TAIL_MIN_TILES=40" "ST355_GEMM_TAIL_MIN_TILES=40 ST355_GEMM_TAIL_KMIN=1024"; do
env $e timeout 500 python bench.py --model sdxl --full --batch 4 --graph --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/g5.json 2> gpurun_out/g5.log
line gpurun_out/g5.json "sdxl-full-b4 [$e]"
done
timeout 1200 python bench.py > gpurun_out/g5_bench_line.json 2> gpurun_out/g5_bench_progress.log
line gpurun_out/g5_bench_line.json "default bench"
