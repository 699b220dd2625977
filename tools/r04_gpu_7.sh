#!/bin/bash
# packed-fp32 softmax VALU in k_attn_fwd64 (tools/kgen/fwd64.py FWD64_PK=1): bit-identity against the committed body (checksums) and rate.
# binaries: tools/attn_lab (committed bodies), tools/attn_lab_fpk6 / _fpk5 (head_dim 128, issue cap 6 / 5), tools/attn_lab_fpk96 (head_dim 96)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/r04_fwd64_packed_fp32.log; : > $out
export LAB_FWD_ONLY=1
run() { echo "-- $1 | $2 | qscale ${3:-1.5} | iters $4" >> $out; LAB_QSCALE=${3:-1.5} LAB_ITERS=$4 timeout 60 tools/$1 $2 2>&1 | grep "forward\|fwd64\|error\|rc=" >> $out; }
for b in attn_lab attn_lab_fpk6 attn_lab_fpk5; do
  run $b "1 8 192 128" 1.5 1; run $b "1 8 1024 128" 8 1; run $b "2 24 4608 128" 8 1; run $b "8 24 4608 128" 1.5 10
done
for b in attn_lab attn_lab_fpk96; do
  run $b "1 16 192 96" 1.5 1; run $b "2 16 4096 96" 8 1; run $b "8 16 4096 96" 1.5 10
done
cat $out
