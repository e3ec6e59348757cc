#!/bin/bash
# Register / spill report of the main Float32 two-stream instances for a set of -D flags.
# Usage: tools/experiments/regs.sh "-DRR_DB=8" ...
cd "$(dirname "$0")/../../rrtmgp.jl_amd/csrc"
for fl in "$@"; do
  echo "== $fl"
  for f in solve_lw solve_sw; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt $fl \
      -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/regs_$$.o 2>&1 \
      | grep -E "Function Name|VGPRs:|Spill|ScratchSize" | paste - - - - - \
      | grep "IfLb1ELb0ELb0ELi[1]" | sed 's/\[-Rpass[^]]*\]//g; s/solve_..\.hip:[0-9]*:1: remark://g; s/  */ /g'
  done
done
rm -f /tmp/regs_$$.o
