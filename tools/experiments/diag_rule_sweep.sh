#!/bin/bash
# Runs on the GPU box (round 5): the occupancy rule of the Float32 one-pass-diagnostic instances (128 VGPRs / 8-layer chunks
# when they admit one more workgroup per CU, up to 64 layers) swept where it was a five-point heuristic: 60 ... 80 layers, with
# and without aerosols.  168 = always the 168-VGPR instances; rule = the shipped rule; force = the 128-VGPR instances whenever
# they admit one more workgroup.  Output gpurun_out/ab_diag_rule.txt (M columns/s, both flux sets; LW / SW ms).
OUT=gpurun_out/ab_diag_rule.txt; mkdir -p gpurun_out; : > $OUT
run() { # label, env, bench args
  env $2 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-legs --clear-sky-diag one-pass --ncol 65536 $3 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-6s %-22s %.3f M col/s   LW %.2f ms  SW %.2f ms' % ('$1', '$3', j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))" >> $OUT 2>&1
}
for nlay in 60 64 72 73 80; do for aer in "" "--aerosols"; do
  args="--nlay $nlay $aer"
  run 168 RRTMGP_HIP_NO_DIAG_HALF=1 "$args"
  run rule X=1 "$args"
  run force RRTMGP_HIP_FORCE_DIAG_HALF=1 "$args"
done; done
cat $OUT
