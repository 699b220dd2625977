#!/bin/bash
# build one attn_lab binary per generator variant:  tools/kgen/variants.sh name1:ENV=val;ENV2=val name2:...   (binaries: tools/attn_lab_<name>)
# a spec whose environment mentions FWD64_ is a variant of the forward body (tools/kgen/fwd64.py), otherwise of the dQ body (tools/kgen/dq64.py)
set -o pipefail
cd /root/repo
mkdir -p simpletuner_amd/csrc/gen/variants
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  if [[ "$envs" == *FWD64_* || "$name" == f* ]]; then mod=fwd64; var=FWD64_OUT; mac=ST355_FWD64_BODY_INC;
  elif [[ "$envs" == *DKV_* || "$name" == k* ]]; then mod=dkv; var=DKV_OUT; mac=ST355_DKV4_BODY_INC;
  else mod=dq64; var=DQ64_OUT; mac=ST355_DQ64_BODY_INC; fi
  ( IFS=';'; for kv in $envs; do export "$kv"; done
    export $var=simpletuner_amd/csrc/gen/variants/${mod}_$name.inc; python -m tools.kgen.$mod 2>/tmp/kgen_err.txt ) || { echo "GENERATOR FAILED: $name: $(tail -1 /tmp/kgen_err.txt)"; rm -f tools/attn_lab_$name; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -D$mac="\"gen/variants/${mod}_$name.inc\"" tools/attn_lab.hip -o tools/attn_lab_$name 2>&1 | grep -E "error" && { echo "BUILD FAILED: $name"; rm -f tools/attn_lab_$name; continue; }
  echo "built tools/attn_lab_$name"
done
