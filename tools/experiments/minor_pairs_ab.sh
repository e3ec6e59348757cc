#!/bin/bash
# Runs on the GPU box: the kminor layout in slot pairs (round 4, later) against the groups of 4 it replaced
# (rrtmgp.jl_amd/variants/r04v3.so = the library of git 4ed9dfd), with and without the bands dealt to the wavefronts by
# slot count.  Usage: tools/experiments/minor_pairs_ab.sh [bench args]; output gpurun_out/ab_minor_pairs.txt
OUT=gpurun_out/ab_minor_pairs.txt; mkdir -p gpurun_out; : > $OUT
run() {  # label, env assignments...
  local label=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-legs ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-28s %.3f M col/s   LW %.2f ms  SW %.2f ms' % ('$label', j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))" >> $OUT 2>&1
}
for rep in 1 2; do
  [ -f rrtmgp.jl_amd/variants/r04v3.so ] && run groups_of_4 RRTMGP_HIP_LIBRARY=$PWD/rrtmgp.jl_amd/variants/r04v3.so
  run pairs_identity_order RRTMGP_HIP_BAND_ORDER=identity
  run pairs_bands_dealt RRTMGP_HIP_BAND_ORDER=dealt
done
cat $OUT
