#!/bin/bash
# round 4, GPU call 1: the (never run) stale-maximum forward in the lab at the bench shape + the dQ-accumulation atomic probe
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export LAB_ITERS=10
timeout 300 tools/attn_lab 8 24 4608 128 > gpurun_out/r04_attn_lab_stale.log 2>&1
timeout 120 tools/probes/atomic_probe > gpurun_out/r04_atomic_probe.log 2>&1
cat gpurun_out/r04_attn_lab_stale.log gpurun_out/r04_atomic_probe.log
