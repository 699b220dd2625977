#!/bin/bash
# usage (on the GPU box): tools/pmc_gemm.sh <impl> <M,N,K> <tag>   -> gpurun_out/pmc_<tag>/*.csv  (separate --pmc passes, no tracing besides kernel-trace)
impl=$1; shape=$2; tag=$3
export TMPDIR=/tmp
R=$PWD
export ST355_GEMM_IMPL=$impl LAB_SHAPE=$shape
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr"; do
  i=$((i+1))
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o p$i --output-format csv -- $R/tools/gemm_lab --child > $out/p$i.log 2>&1)
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "k_gemm" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items():
        print(f"{k:36s} per-dispatch {v / n:16.1f}   (n={n})")
PY
