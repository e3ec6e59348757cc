#!/bin/bash
# Runs on the GPU box: the host-array Layer-2 step (bench.py --l2 fused, and with the clear-sky diagnostic) under different
# ramp maxima of the column pipeline, same session (after the kernels got 10 % faster in round 4 the copy / compute balance
# of the chunks moved).  Output: gpurun_out/l2_ramp_sweep.txt
OUT=gpurun_out/l2_ramp_sweep.txt; mkdir -p gpurun_out; : > $OUT
run() {  # label, env assignment
  for mode in "--l2 fused" "--l2 fused --clear-sky-diag one-pass"; do
    env $2 python bench.py $mode --leg x --steps 10 --warmup 3 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %-40s %.3f M  %.2f ms' % ('$1', '$mode', j['value']/1e6, j['ms_per_step']))" >> $OUT
  done
}
for r in 1 2; do
  run "ramp max 32768 (default)" "X=1"
  run "ramp max 16384" "RRTMGP_HIP_HOST_RAMP_MAX=16384"
  run "ramp max 65536" "RRTMGP_HIP_HOST_RAMP_MAX=65536"
  run "equal 8192" "RRTMGP_HIP_HOST_CHUNK_COLUMNS=8192"
  run "equal 16384" "RRTMGP_HIP_HOST_CHUNK_COLUMNS=16384"
done
cat $OUT
