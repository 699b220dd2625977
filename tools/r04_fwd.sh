#!/bin/bash
# fwd64 vs fwd4 in the lab: small key counts first (prologue / loop / tail paths), then the bench shape.  LABBIN selects the binary.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
BIN=${LABBIN:-tools/attn_lab}
export LAB_ITERS=1
for S in 64 128 192 256 320 512; do
  echo "S=$S $(timeout 60 $BIN 1 8 $S 128 2>&1 | grep "fwd64 vs\|error\|HIP" | tr '\n' '|')"
done
LAB_ITERS=${BENCH_ITERS:-8} timeout 120 $BIN 8 24 4608 128 2>&1 | grep "forward\|fwd64"
