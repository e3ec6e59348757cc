#!/bin/bash
# Builds A/B variants of libhip_rrtmgp.so into rrtmgp.jl_amd/variants/<name>.so (they travel to the GPU box).
# Usage: tools/experiments/build_variants.sh name1="-DFLAG1 -DFLAG2" name2="-DFLAG3" ...
set -e
cd "$(dirname "$0")/../../rrtmgp.jl_amd/csrc"
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  [ "$flags" = "$spec" ] && flags=""
  make -j3 variant NAME=$name EXTRA="$flags" > /tmp/variant_$name.log 2>&1 || { tail -20 /tmp/variant_$name.log; exit 1; }
  echo "built variants/$name.so [$flags]"
done
