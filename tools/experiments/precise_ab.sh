#!/bin/bash
# Runs on the GPU box: the shipped library against the IEEE-Float32 build, same session.  Output gpurun_out/ab_precise.txt
OUT=gpurun_out/ab_precise.txt; mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do
for v in base precise; do
  if [ "$v" = base ]; then unset RRTMGP_HIP_LIBRARY; else export RRTMGP_HIP_LIBRARY=$PWD/rrtmgp.jl_amd/libhip_rrtmgp_precise.so; fi
  python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-legs ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-24s %.3f M col/s   LW %.2f ms  SW %.2f ms' % ('$v', j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))" >> $OUT 2>&1
done; done
cat $OUT
