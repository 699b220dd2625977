#!/bin/bash
# The measurement recipes behind profiles/ as ONE script (run on the GPU box from the repo root, e.g. `gpurun --timeout 2400 -- tools/gpu_lease.sh final`).
# Each recipe writes under gpurun_out/; what is kept is copied into profiles/ by hand (profiles/README.md says which file came from which recipe).
#   final         the end-of-round sequence: rocprofv3 stats + PMC passes (tools/profile_round.sh), pytest -m gpu, smoke(), python bench.py
#   validate      the same without the profile passes
#   gemm_power    GEMM tile-group width sweep: rate, package power / sclk mid-run, FETCH_SIZE per launch        (profiles/r05_gemm_power.md; build the lab binaries first:
#                 for g in 2 4 8 16 32; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DST355_TILE_GROUP=$g tools/gemm_lab.hip -o tools/gemm_lab_g$g; done)
#   tn_slices     split-K slice count of the weight-gradient GEMMs: lab sweep + in-step A/B                      (profiles/r05_tn_slice_count_*)
#   attn_hd96     head_dim-96 attention bodies, committed vs a variant binary tools/attn_lab_<name>              (profiles/r05_attn_lab_hd96_*; tools/kgen/variants.sh builds variants)
#   sd3_stats     rocprofv3 --kernel-trace --stats of the SD3-Medium full fine-tune step by kernel name           (profiles/r05_sd3_full_b8_rocprofv3_kernel_stats.csv)
#   shapes MODEL… per-shape launch table of a bench workload: tools/gpu_lease.sh shapes --model sd3 --full --batch 8   (profiles/r05_*_shapes.txt)
#   two_ranks     bench.py --gpus 2 on one device over gloo (plumbing) + a rank that fails on purpose             (profiles/r05_two_ranks_*)
#   sanity        smoke(), the resume / graph / golden GPU tests, the headline step without secondaries (two minutes)
#   pmc           PMC passes (matrix-pipe busy, LDS, waits) over the labs: GEMM on the step's shapes, attention at the Flux / PixArt 2K shapes   (profiles/r05_pmc_mfma_lds.md)
#   ab NAME ENV=a ENV=b -- bench args…   the same bench command under two environments, back to back            (every same-box A/B in profiles/)
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out
export TMPDIR=/tmp
recipe=$1; shift
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], ":", d["ms_per_step"], "ms/step", d["value"], d["unit"], "frac", d.get("step_frac_of_bf16_mfma_peak"), "vs_baseline", d.get("vs_baseline"), "loss", d.get("loss"))
    print("    ", {k: v["ms_per_step"] for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[1], ": no line (", e, ")")
PY
}
case "$recipe" in
  final|validate)
    t0=$(date +%s)
    [ "$recipe" = final ] && { tools/profile_round.sh ${1:-r05} > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log; }
    timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -15 | cut -c1-300 | tee gpurun_out/gpu_suite_summary.log
    echo "[lease] gpu suite done at +$(( $(date +%s) - t0 )) s" | tee -a gpurun_out/gpu_suite_summary.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "amdgpu.ids" | tail -4 | tee -a gpurun_out/gpu_suite_summary.log
    timeout 900 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_progress.log
    echo "[lease] bench done at +$(( $(date +%s) - t0 )) s"; line gpurun_out/bench_line.json ;;
  gemm_power)
    out=gpurun_out/gemm_tile_order.log; : > $out
    export ST355_GEMM_IMPL=pq
    for shape in 36864,12288,3072 36864,3072,12288 36864,3072,3072; do
      for g in 2 4 8 16 32; do
        it=$(python -c "m,n,k=map(int,'$shape'.split(','));print(int(4.0/(2.0*m*n*k/1.25e15)))")
        (LAB_SHAPE=$shape LAB_ITERS=$it timeout 60 tools/gemm_lab_g$g --child > /tmp/pp_$g.log 2>&1 &)
        sleep 2.6
        p=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Package Power\|Socket Power" | sed 's/.*: //' | tr '\n' ' ')
        while pgrep -x gemm_lab_g$g > /dev/null; do sleep 0.2; done
        r=$(grep -i "tflop" /tmp/pp_$g.log | tail -1)
        d=gpurun_out/pmc_g${g}; rm -rf $d
        (cd /tmp && LAB_SHAPE=$shape LAB_ITERS=4 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OLDPWD/$d -o p --output-format csv -- $OLDPWD/tools/gemm_lab_g$g --child > /dev/null 2>&1)
        f=$(python -c "
import csv, glob
v = [float(r['Counter_Value']) for f in glob.glob('$d/**/*counter_collection.csv', recursive=True) for r in csv.DictReader(open(f)) if 'gemm' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
print('FETCH_SIZE/launch %.3f GB (x2 on gfx950: MI355X_MICROARCH.md)' % (2 * 1024 * sum(v) / max(1, len(v)) / 1e9) if v else 'no counter rows')")
        rm -rf $d
        echo "group $g | $shape | $r | power/sclk: $p | $f" | tee -a $out
      done
    done ;;
  tn_slices)
    L=gpurun_out/tn_slice_count.log; : > $L
    for shape in 6144,1536,32768 1536,6144,32768 4608,1536,32768 1536,1536,32768 3072,3072,36864 1280,1280,16384 640,640,65536 1280,5120,16384; do
      for ks in -1 0 3 5 7 9; do ST355_TN_KS=$ks LAB_TN=$shape timeout 60 tools/gemm_lab --child 2>&1 | grep "TN" >> $L; done
    done
    cut -c1-150 $L
    for m in -1 0 -1 0; do ST355_TN_KS=$m timeout 400 python bench.py --model sd3 --full --batch 8 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/sd3_full_b8_tnks$m.json 2> /dev/null; line gpurun_out/sd3_full_b8_tnks$m.json; done ;;
  attn_hd96)
    L=gpurun_out/attn_lab_hd96.log; : > $L
    for v in "$@"; do
      echo "=== $v (LAB_DVALID=72, B1 H16 S16384 d96)" >> $L
      LAB_DVALID=72 LAB_ITERS=6 timeout 120 tools/attn_lab_$v 1 16 16384 96 2>&1 | grep -v "generation 1\|row-major" >> $L
      for sh in "2 8 4096" "2 4 1000" "1 2 64" "1 2 192"; do LAB_DVALID=72 LAB_ITERS=3 timeout 120 tools/attn_lab_$v $sh 96 2>&1 | grep -i "mismatch" >> $L; done
    done
    grep -i "TFLOP\|===\|MISMATCH\|identical\|rounding" $L | cut -c1-160 ;;
  sd3_stats)
    R=$PWD; out=$R/gpurun_out/prof_sd3; rm -rf $out; mkdir -p $out
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- python $R/bench.py --model sd3 --full --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench.log 2>&1)
    python - <<'PY'
import csv, glob
rows = [r for f in glob.glob("gpurun_out/prof_sd3/stats/**/*kernel_stats.csv", recursive=True) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/sd3_full_b8_rocprofv3_kernel_stats.csv", "w") as g:
    g.write("kernel,calls,total_ms,avg_us,pct\n")
    for r in rows:
        g.write(f"\"{r['Name'].split('(')[0][:110]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.2f},{100 * float(r['TotalDurationNs']) / tot:.2f}\n")
for r in rows[:40]:
    print(f"{r['Name'].split('(')[0][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6 / 4:9.3f} ms/step {float(r['AverageNs']) / 1e3:9.1f} us")
PY
    rm -rf gpurun_out/prof_sd3/stats ;;
  shapes)
    timeout 400 python bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --prof-dump gpurun_out/shapes_dump.csv > gpurun_out/shapes_line.json 2> gpurun_out/shapes.log
    python tools/prof_shapes.py gpurun_out/shapes_dump.csv 3 | tee gpurun_out/shapes.txt | head -70 | cut -c1-150; rm -f gpurun_out/shapes_dump.csv; line gpurun_out/shapes_line.json ;;
  two_ranks)
    export ST355_BENCH_SHARE_GPU=1
    for extra in "--rank 16" "--full"; do
      timeout 300 python bench.py --gpus 2 --model sd3 $extra --batch 1 --layers 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/two_ranks_$(echo $extra | tr -d ' -').json 2> gpurun_out/two_ranks.log
      echo "exit code $?"; line gpurun_out/two_ranks_$(echo $extra | tr -d ' -').json
    done
    ST355_BENCH_FAIL_RANK=1 timeout 200 python bench.py --gpus 2 --model sd3 --rank 16 --batch 1 --layers 2 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/two_ranks_one_fails.log
    echo "failing-rank run: exit code $? (non-zero expected)"; grep "FAILED" gpurun_out/two_ranks_one_fails.log | head -2 | cut -c1-300 ;;
  ab)
    name=$1; shift; envs=()
    while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done; shift
    for rep in 1 2; do for e in "${envs[@]}"; do
      env "$e" timeout 500 python bench.py "$@" --no-cpu-baseline > gpurun_out/ab_${name}_${e//[^A-Za-z0-9]/_}.json 2> gpurun_out/ab_${name}.log; echo "[$e]"; line gpurun_out/ab_${name}_${e//[^A-Za-z0-9]/_}.json
    done; done ;;
  sanity)       # two minutes: smoke(), the resume / graph / golden GPU tests, the headline step without its secondaries
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "amdgpu.ids" | tail -3
    timeout 600 python -m pytest tests/test_optimizer_state_gpu.py tests/test_trainer_graph_gpu.py tests/test_golden_gpu.py tests/test_adamw_bf16_gpu.py -q 2>&1 | tail -2
    timeout 300 python bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/sanity_line.json 2> /dev/null; line gpurun_out/sanity_line.json ;;
  pmc)          # matrix-pipe busy / LDS activity / wait fraction by PMC (counters only + --kernel-trace): the GEMM on the step's shapes, attention at the Flux and PixArt 2K shapes
    for sh in 36864,12288,3072 36864,3072,12288 36864,3072,3072; do tools/gemm_pmc.sh r05_$(echo $sh | tr ',' 'x') $sh | cut -c1-330; done
    tools/attn_pmc.sh r05_flux | cut -c1-330
    ATTN_SHAPE="1 16 16384 96" LAB_DVALID=72 tools/attn_pmc.sh r05_pixart | cut -c1-330
    cat gpurun_out/gemm_pmc_r05_*/summary.txt gpurun_out/attn_pmc_r05_*/summary.txt > gpurun_out/r05_pmc_mfma_lds_raw.txt; rm -rf gpurun_out/gemm_pmc_r05_*/p[12] gpurun_out/attn_pmc_r05_*/p[12] ;;
  *) echo "usage: tools/gpu_lease.sh final|validate|gemm_power|tn_slices|attn_hd96 NAMES…|sd3_stats|shapes BENCH_ARGS…|two_ranks|ab NAME ENV=a ENV=b -- BENCH_ARGS…"; exit 2 ;;
esac
