#!/bin/bash
# Runs on the GPU box: bench.py once per library variant, same session (box-to-box variation is +-3 %).
# Usage: tools/experiments/ab.sh <tag> [variant ...]     ("base" = the shipped library); output gpurun_out/ab_<tag>.txt
TAG=$1; shift
OUT=gpurun_out/ab_$TAG.txt; mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset RRTMGP_HIP_LIBRARY; else export RRTMGP_HIP_LIBRARY=$PWD/rrtmgp.jl_amd/variants/$v.so; fi
  python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-legs ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-24s %.3f M col/s   LW %.2f ms  SW %.2f ms' % ('$v', j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))" >> $OUT 2>&1
done; done
cat $OUT
