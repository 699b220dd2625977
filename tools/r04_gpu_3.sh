#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export LAB_ITERS=1
for S in 64 128 192 256 512; do
  echo "== S=$S" >> gpurun_out/r04_dq64_sizes.log
  timeout 60 tools/attn_lab 1 8 $S 128 2>&1 | grep "dQ (dq64\|rc=\|error\|HIP" >> gpurun_out/r04_dq64_sizes.log
done
cat gpurun_out/r04_dq64_sizes.log
