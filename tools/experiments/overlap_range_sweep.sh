#!/bin/bash
# Runs on the GPU box (round 5): where do two lanes pay?  Fused step on device arrays (72 layers, aerosols), one lane
# (RRTMGP_HIP_STEP_OVERLAP=0) against two (=1), by shard size.  Output gpurun_out/ab_overlap_range.txt
OUT=gpurun_out/ab_overlap_range.txt; : > $OUT
export NCOLS=${NCOLS:-256,512,768,1024,1536,2048,3072,4096,6144,8192,12288,16384,24576}
for rep in 1 2; do for o in 0 1; do
  echo "== overlap $o" >> $OUT
  RRTMGP_HIP_STEP_OVERLAP=$o python tools/experiments/small_step_host_cost.py 2>&1 | grep "^ncol" | cut -c1-36 >> $OUT
done; done
cat $OUT
