#!/bin/bash
# tenth GPU-box session of round 3: rocprofv3 --kernel-trace --stats of the Flux.1-dev full-rank step (kernel-level evidence behind the flux_full_rank secondary)
tag=${1:-r03r}
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp
timeout 110 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $R/bench.py --model flux --full --batch 8 --steps 2 --warmup 1 --optimizer adamw_bf16 \
  --gradient-checkpointing --ckpt-interval 3 --ckpt-stride 4 --no-cpu-baseline --no-secondary > $out/bench_stats.log 2>&1
cd $R
f=$(ls $out/stats/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp $f gpurun_out/${tag}_flux_full_rank_rocprofv3_kernel_stats_raw.csv; head -25 $f | cut -c1-160; fi
rm -f $out/stats/*kernel_trace.csv
tail -2 $out/bench_stats.log | cut -c1-300
