#!/bin/bash
# Runs on the GPU box (round 5): do LW and SW workgroups help each other when they SHARE a CU?  LW is the vector-memory-heavy
# kernel (pipeline 73 % busy, VALU 45 %), SW the VALU-heavy one (59 % / 65 %).  A persistent grid fills every resident slot, so
# two streams alone never co-reside at large ncol; capping each grid at 2 workgroups per CU and launching on two streams does.
# Output gpurun_out/ab_mixed.txt: M columns/s of the default workload.
OUT=gpurun_out/ab_mixed.txt; mkdir -p gpurun_out; : > $OUT
run() { # label, env, args
  python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-legs $3 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-40s %.3f M col/s   %.2f ms/step   LW %.2f ms  SW %.2f ms' % ('$1', j['value'] / 1e6, j['ms_per_step'], j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))" >> $OUT 2>&1
}
for rep in 1 2; do
  run "one stream, 4 wg/cu (shipped)" "" ""
  run "two streams, 4 wg/cu each" "" "--streams 2"
  RRTMGP_HIP_MAX_WG_PER_CU=2 run "two streams, 2 wg/cu each (co-resident)" "" "--streams 2"
  RRTMGP_HIP_MAX_WG_PER_CU=2 run "one stream, 2 wg/cu" "" ""
  RRTMGP_HIP_MAX_WG_PER_CU=3 run "two streams, 3 wg/cu each" "" "--streams 2"
done
cat $OUT
