#!/bin/bash
# usage (on the GPU box, repo root): tools/gemm_pmc.sh <tag> M,N,K   — rocprofv3 PMC passes (counters only) over tools/gemm_lab on one shape
tag=${1:-r02}; shape=${2:-36864,12288,3072}
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/gemm_pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp
export ST355_GEMM_IMPL=pq LAB_SHAPE=$shape LAB_ITERS=6
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/p1 -o p1 --output-format csv -- $R/tools/gemm_lab --child > $out/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $out/p2 -o p2 --output-format csv -- $R/tools/gemm_lab --child > $out/p2.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
out = "$out"
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fo:
    for k, cs in per.items():
        if "gemm" not in k: continue
        m = {c: sum(v.values()) / len(v) for c, v in cs.items()}
        mf = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0; el = m.get("GRBM_GUI_ACTIVE", 1) / 8.0
        line = ("$shape " + k + "  MFMA_busy=%.3f  LDS_active=%.3f  bank_conflict/lds_active=%.4f  wait/wave_cycles=%.3f  " %
                (mf / el, m.get("SQ_LDS_IDX_ACTIVE", 0) / 256.0 / el, m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1)),
                 m.get("SQ_WAIT_INST_ANY", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1))) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(m.items())))
        print(line); fo.write(line + "\\n")
PY
