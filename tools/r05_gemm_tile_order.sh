#!/bin/bash
# Tile-order experiment for the GEMM class (next round; DESIGN.md §7 "order of work" item 1): the persistent / one-tile kernels walk tiles in groups of ST355_TILE_GROUP
# m-tiles per W panel; with 32 concurrent tiles per XCD the block shape GROUP x (32 / GROUP) sets how many operand panels an XCD fetches per round
# (8 x 4: 12 panels; 4 x 8: 12; 16 x 2: 18; 2 x 16: 18) and which of them neighbouring XCDs share through the Infinity Cache.  For each width: rate with random
# operands (the power-limited figure) and FETCH_SIZE per launch.
#   build here (CPU container):  tools/r05_gemm_tile_order.sh build       -> tools/gemm_lab_g<width>
#   run on the GPU box:          gpurun --timeout 900 -- tools/r05_gemm_tile_order.sh run
set -o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
WIDTHS="2 4 8 16 32"
if [ "$1" = "build" ]; then
  for g in $WIDTHS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DST355_TILE_GROUP=$g tools/gemm_lab.hip -o tools/gemm_lab_g$g 2>&1 | grep -E "error" && echo "BUILD FAILED g$g"
  done
  ls -la tools/gemm_lab_g*
  exit 0
fi
mkdir -p gpurun_out; out=gpurun_out/r05_gemm_tile_order.log; : > $out
export TMPDIR=/tmp ST355_GEMM_IMPL=pq LAB_ITERS=10
for shape in 36864,12288,3072 36864,3072,12288 36864,3072,3072; do
  for g in $WIDTHS; do
    r=$(LAB_SHAPE=$shape timeout 120 tools/gemm_lab_g$g --child 2>&1 | grep -i "tflop" | tail -1)
    d=gpurun_out/r05_tile_order_pmc_g${g}_$(echo $shape | tr ',' 'x'); rm -rf $d
    (cd /tmp && LAB_SHAPE=$shape LAB_ITERS=4 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OLDPWD/$d -o p --output-format csv -- $OLDPWD/tools/gemm_lab_g$g --child > /dev/null 2>&1)
    f=$(python - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("$d/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("FETCH_SIZE/launch %.3f GB (x2 on gfx950: MI355X_MICROARCH.md)" % (2 * 1024 * sum(v) / max(1, len(v)) / 1e9) if v else "no counter rows")
PY
)
    echo "group $g | $shape | $r | $f" | tee -a $out
  done
done
