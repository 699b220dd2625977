#!/bin/bash
# dq64 vs dq bit check at small key counts (1..8 tiles) — isolates prologue / loop / drain paths.  LABBIN selects the lab binary.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export LAB_ITERS=1
BIN=${LABBIN:-tools/attn_lab}
for S in 64 128 192 256 320 512; do
  echo "S=$S $(timeout 60 $BIN 1 8 $S 128 2>&1 | grep "dQ (dq64\|error\|HIP" | sed 's/.*path: //')"
done
