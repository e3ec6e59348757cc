#!/bin/bash
# Runs on the GPU box (round 5): short fused steps (512 ... 4096 columns x 72, aerosols, device arrays): previous commit
# (variants/prevq.so: memsets in front of the kernels, LW queued first, net sums on the main lane) against the working tree with
# RRTMGP_HIP_STEP_ORDER = 0 (same order, self-resetting queue only), 1 (+ net sums on the second lane), 2 (+ SW queued first).
OUT=gpurun_out/ab_small_step.txt; : > $OUT
for rep in 1 2; do
  for v in prevq order0 order1 order2; do
    if [ "$v" = prevq ]; then export RRTMGP_HIP_LIBRARY=$PWD/rrtmgp.jl_amd/variants/prevq.so; unset RRTMGP_HIP_STEP_ORDER
    else unset RRTMGP_HIP_LIBRARY; export RRTMGP_HIP_STEP_ORDER=${v#order}; fi
    echo "== $v" >> $OUT
    python tools/experiments/small_step_host_cost.py 2>&1 | grep "^ncol" | cut -c1-40 >> $OUT
  done
done
cat $OUT
