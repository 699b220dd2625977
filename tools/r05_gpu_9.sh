#!/bin/bash
# round 5, GPU call 9: the head_dim-96 dK/dV body with 5 contraction k-steps and the fragment-ring position carried across phases, against the committed body
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r05_attn_lab_hd96_dkv_ring.log; : > $L
for v in h96base k96ring; do
  echo "=== $v (LAB_DVALID=72, B1 H16 S16384 d96)" >> $L
  LAB_DVALID=72 LAB_ITERS=6 timeout 120 tools/attn_lab_$v 1 16 16384 96 2>&1 | grep "dkv4 + dq64\|MISMATCH\|identical\|rounding" >> $L
done
echo "=== k96ring at SD 1.5's padded 80 (LAB_DVALID=80, B2 H8 S4096 d96), a ragged S (B2 H4 S1000), one tile (S64), three tiles (S192)" >> $L
LAB_DVALID=80 LAB_ITERS=3 timeout 120 tools/attn_lab_k96ring 2 8 4096 96 2>&1 | grep -i "mismatch\|identical\|rounding" >> $L
LAB_DVALID=72 LAB_ITERS=3 timeout 120 tools/attn_lab_k96ring 2 4 1000 96 2>&1 | grep -i "mismatch\|identical\|rounding" >> $L
LAB_DVALID=72 LAB_ITERS=3 timeout 120 tools/attn_lab_k96ring 1 2 64 96 2>&1 | grep -i "mismatch\|identical\|rounding" >> $L
LAB_DVALID=72 LAB_ITERS=3 timeout 120 tools/attn_lab_k96ring 1 2 192 96 2>&1 | grep -i "mismatch\|identical\|rounding" >> $L
cat $L | cut -c1-170
