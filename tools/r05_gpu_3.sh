#!/bin/bash
# round 5, GPU call 3: per-shape launch tables (bench.py --prof-dump) of the SD3-Medium full fine-tune step (b8, 1024^2) and the mixed-bucket run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --model sd3 --full --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --prof-dump gpurun_out/r05_sd3_full_b8_dump.csv > gpurun_out/r05_sd3_full_b8_line.json 2> gpurun_out/r05_sd3_full_b8.log
python tools/prof_shapes.py gpurun_out/r05_sd3_full_b8_dump.csv 4 > gpurun_out/r05_sd3_full_b8_shapes.txt; head -70 gpurun_out/r05_sd3_full_b8_shapes.txt
rm -f gpurun_out/r05_sd3_full_b8_dump.csv
timeout 300 python bench.py --model sd3 --rank 128 --batch 3 --steps 4 --warmup 2 --no-cpu-baseline --prof-dump gpurun_out/r05_sd3_r128_dump.csv > gpurun_out/r05_sd3_r128_line.json 2> gpurun_out/r05_sd3_r128.log
python tools/prof_shapes.py gpurun_out/r05_sd3_r128_dump.csv 4 > gpurun_out/r05_sd3_r128_shapes.txt; head -40 gpurun_out/r05_sd3_r128_shapes.txt
rm -f gpurun_out/r05_sd3_r128_dump.csv
