#!/bin/bash
# round 5, GPU call 13: the N > 1 code path of bench.py on the 1-GPU box after this round's changes (guarded main, collective timeouts): two ranks on one device over gloo
# (RCCL refuses two ranks per device) — plumbing evidence, not a throughput number; plus a rank that fails on purpose must take the job down with a non-zero exit code
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export ST355_BENCH_SHARE_GPU=1
timeout 300 python bench.py --gpus 2 --model sd3 --rank 16 --batch 1 --layers 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05_two_ranks_shared_gpu_gloo_bench_line.json 2> gpurun_out/r05_two_ranks_shared_gpu_gloo.log
echo "exit code $?"; tail -c 600 gpurun_out/r05_two_ranks_shared_gpu_gloo_bench_line.json; echo
timeout 300 python bench.py --gpus 2 --model sd3 --full --batch 1 --layers 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05_two_ranks_shared_gpu_gloo_full_bench_line.json 2> gpurun_out/r05_two_ranks_shared_gpu_gloo_full.log
echo "exit code $?"; python - <<'PY'
import json
for f in ("r05_two_ranks_shared_gpu_gloo_bench_line.json", "r05_two_ranks_shared_gpu_gloo_full_bench_line.json"):
    try:
        d = json.loads(open("gpurun_out/" + f).read().strip().splitlines()[-1])
        print(f, "n_gpus", d["n_gpus"], d["ms_per_step"], "ms/step; comm:", {k: d["comm"].get(k) for k in ("path", "mode", "fp32_reduce", "buckets", "overlap_frac", "exposed_tail_ms")})
    except Exception as e:
        print(f, "no line:", e)
PY
# a failing rank: rank 1 raises inside its first step (ST355_BENCH_FAIL_RANK, lab hook) -> the launcher must stop rank 0 and the command must exit non-zero, promptly
ST355_BENCH_FAIL_RANK=1 timeout 200 python bench.py --gpus 2 --model sd3 --rank 16 --batch 1 --layers 2 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r05_two_ranks_one_fails.log
echo "failing-rank run: exit code $? (non-zero expected)"; grep "FAILED" gpurun_out/r05_two_ranks_one_fails.log | head -2 | cut -c1-300
