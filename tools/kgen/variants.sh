#!/bin/bash
# build one attn_lab binary per generator variant:  tools/kgen/variants.sh name1:ENV=val,ENV2=val name2:...   (binaries: tools/attn_lab_<name>)
set -o pipefail
cd /root/repo
mkdir -p simpletuner_amd/csrc/gen/variants
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  ( IFS=';'; for kv in $envs; do export "$kv"; done
    DQ64_OUT=simpletuner_amd/csrc/gen/variants/dq64_$name.inc python -m tools.kgen.dq64 2>/tmp/kgen_err.txt ) || { echo "GENERATOR FAILED: $name: $(tail -1 /tmp/kgen_err.txt)"; rm -f tools/attn_lab_$name; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DST355_DQ64_BODY_INC="\"gen/variants/dq64_$name.inc\"" tools/attn_lab.hip -o tools/attn_lab_$name 2>&1 | grep -E "error" && { echo "BUILD FAILED: $name"; rm -f tools/attn_lab_$name; continue; }
  echo "built tools/attn_lab_$name"
done
