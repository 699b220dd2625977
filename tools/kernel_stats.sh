R=$PWD
for cfg in "sdxl --model sdxl --lora --rank 16 --batch 16" "sd3 --model sd3 --full --batch 8 --buckets"; do
  set -- $cfg; name=$1; shift
  out=$R/gpurun_out/prof_$name; rm -rf $out; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $R/bench.py "$@" --steps 4 --warmup 1 --no-cpu-baseline --no-prof > $out/bench.log 2>&1)
  NAME=$name python - <<'PY'
import csv, glob, os
name = os.environ["NAME"]
rows = [r for f in glob.glob(f"gpurun_out/prof_{name}/stats/**/*kernel_stats.csv", recursive=True) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(f"gpurun_out/r06_{name}_rocprofv3_kernel_stats.csv", "w") as g:
    g.write("kernel,calls,total_ms,avg_us,pct\n")
    for r in rows:
        g.write(f"\"{r['Name'].split('(')[0][:110]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.2f},{100 * float(r['TotalDurationNs']) / tot:.2f}\n")
print(name, "total ms over 5 steps", tot / 1e6)
for r in rows[:45]:
    print(f"{r['Name'].split('(')[0][:80]:80s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6 / 5:9.3f} ms/step {float(r['AverageNs']) / 1e3:9.1f} us")
PY
  rm -rf $out/stats
done
