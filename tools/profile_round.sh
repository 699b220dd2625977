#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag> [bench args…]     e.g. r01b   |   r06_sdxl_lora --model sdxl --lora --rank 16 --batch 16
# (extra arguments replace the default command's `--no-secondary`: the profile is then of THAT workload, eager launches)
# 1) rocprofv3 --kernel-trace --stats of the default bench command        -> gpurun_out/prof_<tag>/stats_*.csv
# 2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; no other tracing) -> gpurun_out/prof_<tag>/pmc_*.csv
# then tools/profile_summarise.py turns them into profiles/<tag>_kernel_stats.csv and profiles/<tag>_hbm_traffic.json
tag=${1:-r01b}; shift
extra="--no-secondary"; [ $# -gt 0 ] && extra="$*"
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra > $out/bench_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $extra --no-prof > $out/bench_pmc_$c.log 2>&1
done
cd $R
python tools/profile_summarise.py $out $tag
cp profiles/${tag}_* $R/gpurun_out/ 2>/dev/null
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE      # the raw traces (hundreds of MB for the UNet steps) stay on the box: gpurun merges at most 64 MiB back
