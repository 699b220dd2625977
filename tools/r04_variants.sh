#!/bin/bash
# run generator-variant lab binaries: tools/r04_variants.sh "<B H S d>" name1 name2 ...  -> gpurun_out/r04_variants.log (dq64 timing + bit check per variant)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
shape=$1; shift
export LAB_ITERS=${LAB_ITERS:-8}
for n in "$@"; do
  r=$(timeout 100 tools/attn_lab_$n $shape 2>&1 | grep "dq64 *attn_bwd_dq\|dQ (dq64")
  echo "$n | $(echo "$r" | grep attn_bwd_dq | awk '{print $7, $8, $9, $10}') | $(echo "$r" | grep "dQ (dq64" | sed 's/.*path: //')" | tee -a gpurun_out/r04_variants.log
done
