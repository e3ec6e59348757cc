#!/bin/bash
# Runs on the GPU box (round 5): main Float32 instances, 8-layer chunks (4 workgroups per CU) against 16-layer chunks (3 per CU)
# at 72 ... 96 layers, no aerosols.  forced = RRTMGP_HIP_FORCE_HALF_CHUNKS=1 (whenever the 8-layer records fit a quarter of the LDS).
OUT=gpurun_out/ab_half_rule.txt; : > $OUT
for nlay in 72 80 84 88 96; do for e in RRTMGP_HIP_NO_HALF_CHUNKS=1 X=1 RRTMGP_HIP_FORCE_HALF_CHUNKS=1; do
  env $e python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-legs --ncol 65536 --nlay $nlay 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-34s nlay %3d  %.3f M col/s  LW %.2f ms  SW %.2f ms' % ('$e', $nlay, j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))" >> $OUT
done; done
cat $OUT
