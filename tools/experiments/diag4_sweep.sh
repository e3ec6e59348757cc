#!/bin/bash
# Runs on the GPU box: the Float32 one-pass-diagnostic instances at 128 VGPRs / 8-layer chunks (taken when they admit one more
# resident workgroup than the 168-VGPR instances) against RRTMGP_HIP_NO_DIAG_HALF=1 (always the 168-VGPR instances), across
# workloads, same session.  Output: gpurun_out/ab_diag4_sweep.txt
OUT=gpurun_out/ab_diag4_sweep.txt; mkdir -p gpurun_out; : > $OUT
run() { # label, env, bench args
  env $2 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-legs --clear-sky-diag one-pass $3 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-14s %-30s %.3f M col/s   LW %.2f ms  SW %.2f ms' % ('$1', '$3', j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))" >> $OUT 2>&1
}
for args in "" "--cld-frac 0.5" "--aerosols" "--nlay 72 --ncol 65536" "--nlay 96 --ncol 65536" "--nlay 128 --ncol 32768" "--nlay 40 --ncol 131072" "--dtype f64 --ncol 65536"; do
  run "168 VGPR only" RRTMGP_HIP_NO_DIAG_HALF=1 "$args"
  run "by occupancy" X=1 "$args"
  run "by occupancy" X=1 "$args"
  run "168 VGPR only" RRTMGP_HIP_NO_DIAG_HALF=1 "$args"
done
cat $OUT
