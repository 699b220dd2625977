#!/bin/bash
# run dK/dV-variant lab binaries: tools/r04_kvariants.sh "<B H S d>" name1 name2 ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
shape=$1; shift
export LAB_ITERS=${LAB_ITERS:-6}
for n in "$@"; do
  r=$(timeout 100 tools/attn_lab_$n $shape 2>&1 | grep "dkv4")
  echo "$n | $(echo "$r" | grep "dkv4 + dq64 *attn_bwd_dkv" | awk '{print $7, $8, $9}') | $(echo "$r" | grep "dK, dkv4" | sed 's/.*dkv3: //')" | tee -a gpurun_out/r04_kvariants.log
done
