#!/bin/bash
# usage (on the GPU box, repo root): tools/attn_pmc.sh <tag>   — rocprofv3 PMC passes (counters only, --kernel-trace) over tools/attn_lab at the Flux shape
tag=${1:-r02}
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/attn_pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp
export ATTN_LAB_CHILD=1 ST355_ATTN_FWD=${ATTN_GEN:-4} LAB_ITERS=4      # ATTN_GEN=1: the child that also runs both backward forms
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/p1 -o p1 --output-format csv -- $R/tools/attn_lab ${ATTN_SHAPE:-8 24 4608 128} > $out/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $out/p2 -o p2 --output-format csv -- $R/tools/attn_lab ${ATTN_SHAPE:-8 24 4608 128} > $out/p2.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
out = "$out"
# one row per (dispatch, counter instance): sum the instances of a dispatch, then average over the dispatches of that kernel in that pass
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fo:
    for k, cs in per.items():
        if "attn" not in k: continue
        m = {c: sum(v.values()) / len(v) for c, v in cs.items()}
        mf = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0; el = m.get("GRBM_GUI_ACTIVE", 1) / 8.0
        line = (k + "  MFMA_busy_per_SIMD/elapsed_per_XCD=%.3f  LDS_idx_active_per_CU/elapsed=%.3f  bank_conflict/lds_active=%.4f  wait/wave_cycles=%.3f  " %
                (mf / el, m.get("SQ_LDS_IDX_ACTIVE", 0) / 256.0 / el, m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1)),
                 m.get("SQ_WAIT_INST_ANY", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1))) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(m.items())))
        print(line); fo.write(line + "\n")
PY
