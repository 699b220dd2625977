#!/bin/bash
# usage: tools/kasm.sh <file.hip> <mangled-kernel-substring>  -> /tmp/kasm.s (the kernel's gfx950 assembly)
set -e
cd /root/repo/simpletuner_amd/csrc
rm -rf /tmp/kasm_dir && mkdir -p /tmp/kasm_dir
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=fast -c "$1" -o /tmp/kasm_dir/out.o -save-temps=obj 2>/dev/null
S=$(ls /tmp/kasm_dir/*-hip-amdgcn-amd-amdhsa-gfx950.s)
awk -v k="$2" 'index($0, k) && /^_Z[A-Za-z0-9_]*:/ {p=1} p {print} p && /s_endpgm/ {exit}' "$S" > /tmp/kasm.s
wc -l /tmp/kasm.s
