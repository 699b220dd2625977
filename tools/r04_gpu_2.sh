#!/bin/bash
# round 4, GPU call 2: first run of the hand-scheduled dq64 kernel in the lab (bit-compare with k_attn_bwd_dq, timing)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export LAB_ITERS=${LAB_ITERS:-5}
timeout 120 tools/attn_lab 2 24 4608 128 > gpurun_out/r04_attn_lab_dq64_b2.log 2>&1; echo "rc=$?" >> gpurun_out/r04_attn_lab_dq64_b2.log
grep -v "forward" gpurun_out/r04_attn_lab_dq64_b2.log
timeout 120 tools/attn_lab 8 24 4608 128 > gpurun_out/r04_attn_lab_dq64_b8.log 2>&1; echo "rc=$?" >> gpurun_out/r04_attn_lab_dq64_b8.log
grep -v "forward" gpurun_out/r04_attn_lab_dq64_b8.log
