#!/bin/bash
# round 5, end-of-round sequence on one MI355X (what the driver runs, plus the rocprofv3 passes the bench line's `traffic` cites):
#   1) tools/profile_round.sh r05  — rocprofv3 --kernel-trace --stats of the default Flux command + separate --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/r05_*
#   2) python -m pytest tests -m gpu
#   3) __graft_entry__.smoke()
#   4) python bench.py   (the default line with its secondaries and parity legs)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
echo "(profile passes: profiles/r05_* of the earlier lease)"
echo "[final] profile passes done at +$(( $(date +%s) - t0 )) s"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -15 | cut -c1-300 | tee gpurun_out/r05_gpu_suite_summary_3.log
echo "[final] gpu suite done at +$(( $(date +%s) - t0 )) s" | tee -a gpurun_out/r05_gpu_suite_summary_3.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee -a gpurun_out/r05_gpu_suite_summary_3.log
timeout 900 python bench.py > gpurun_out/r05zzz_bench_line.json 2> gpurun_out/r05zzz_bench_progress.log
echo "[final] bench done at +$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05zzz_bench_line.json").read().strip().splitlines()[-1])
print("headline:", d["value"], d["unit"], d["ms_per_step"], "ms/step; roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch")})
for n, s in d["secondary"].items():
    print(" ", n, {k: s.get(k) for k in ("value", "ms_per_step", "step_frac_of_bf16_mfma_peak", "vs_baseline", "error")}, s.get("same_row_under_fp32_adamw"))
for n, s in d.get("parity_at_other_configs", {}).items():
    print(" ", n, {k: s.get(k) for k in ("pred_rel_l2", "grad_worst_rel_l2", "grad_worst_vs_its_tolerance", "seconds", "error", "skipped")})
print("  parity_at_config:", {k: d["parity_at_config"].get(k) for k in list(d["parity_at_config"])[:8]} if d.get("parity_at_config") else None)
PY
