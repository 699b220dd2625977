#!/bin/bash
# round 5, GPU call 14: rocprofv3 --kernel-trace --stats of the SD3-Medium full fine-tune step (batch 8): which kernels make up its `elementwise` class (19 ms, 737 launches)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/prof_r05_sd3; rm -rf $out; mkdir -p $out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $R/bench.py --model sd3 --full --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench.log 2>&1
cd $R
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/prof_r05_sd3/stats/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/r05_sd3_full_b8_rocprofv3_kernel_stats.csv", "w") as g:
    g.write("kernel,calls,total_ms,avg_us,pct\n")
    for r in rows:
        g.write(f"\"{r['Name'].split('(')[0][:110]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.2f},{100 * float(r['TotalDurationNs']) / tot:.2f}\n")
for r in rows[:45]:
    print(f"{r['Name'].split('(')[0][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6 / 4:9.3f} ms/step {float(r['AverageNs']) / 1e3:9.1f} us")
PY
rm -rf gpurun_out/prof_r05_sd3/stats
