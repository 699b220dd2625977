#!/bin/bash
# usage (GPU box): tools/power_probe.sh <binary> [impl]  -> sustained TFLOP/s of the 8192^3 GEMM with the package power / sclk sampled mid-run
bin=$1; impl=${2:-pq}
(LAB_ITERS=3000 LAB_SHAPE=8192,8192,8192 ST355_GEMM_IMPL=$impl $bin --child > /tmp/pp.log 2>&1 &)
sleep 1.6
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/.*: //' | tr '\n' ' '
echo
while pgrep -x $(basename $bin) > /dev/null; do sleep 0.2; done
grep TFLOP /tmp/pp.log
