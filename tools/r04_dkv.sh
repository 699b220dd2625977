#!/bin/bash
# dkv4 vs dkv3 in the lab: small sizes (1, 2, 3, 4 tiles; ragged), then the bench shape
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
BIN=${LABBIN:-tools/attn_lab}
export LAB_ITERS=1
for S in 64 128 192 256 300 512; do
  echo "S=$S $(timeout 60 $BIN 1 8 $S 128 2>&1 | grep "dkv4 vs\|error\|HIP\|rc=" | tr '\n' '|')"
done
LAB_ITERS=${BENCH_ITERS:-6} timeout 120 $BIN 8 24 4608 128 2>&1 | grep "bwd\|dkv4"
