#!/bin/bash
# fourth GPU-box session of round 3: the C VAE encoder against the per-kernel sequencing, a rocprofv3 kernel-stats pass of the SD3 full fine-tune step
tag=${1:-r03e}
mkdir -p gpurun_out
R=$PWD
timeout 600 python -m pytest tests/test_vae_gpu.py tests/test_cache_feed_gpu.py tests/test_prepare_batch_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/${tag}_pytest_vae.log 2>&1
tail -4 gpurun_out/${tag}_pytest_vae.log
timeout 300 python bench.py --model vae --no-cpu-baseline > gpurun_out/${tag}_vae_bench.json 2> gpurun_out/${tag}_vae_bench.err; tail -c 400 gpurun_out/${tag}_vae_bench.json; echo
export TMPDIR=/tmp
out=$R/gpurun_out/prof_${tag}_sd3
rm -rf $out; mkdir -p $out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $R/bench.py --model sd3 --full --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_stats.log 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("$out/stats/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0]))) if f else []
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
cp $out/stats/*kernel_stats.csv gpurun_out/${tag}_sd3_full_rocprofv3_kernel_stats_raw.csv 2>/dev/null
