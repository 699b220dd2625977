#!/bin/bash
# run forward-variant lab binaries: tools/r04_fvariants.sh "<B H S d>" name1 name2 ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
shape=$1; shift
export LAB_ITERS=${LAB_ITERS:-8}
for n in "$@"; do
  r=$(LAB_FWD_TRACE=$LAB_FWD_TRACE timeout 100 tools/attn_lab_$n $shape 2>&1 | grep "fwd64")
  echo "$n | $(echo "$r" | grep "forward, fwd64" | awk '{print $5, $6, $7, $8}') | $(echo "$r" | grep "O, fwd64" | sed 's/.*fwd4: //') | $(echo "$r" | grep "trace wave 0" | sed 's/.*wave 0: //')" | tee -a gpurun_out/r04_fvariants.log
done
