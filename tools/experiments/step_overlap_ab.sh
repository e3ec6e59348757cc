#!/bin/bash
# Runs on the GPU box: the device-resident Layer-2 step (bench.py --fused-step) with the SW kernels on the workspace's
# second lane (RRTMGP_HIP_STEP_OVERLAP=1) against one lane, by batch size.
mkdir -p gpurun_out; out=gpurun_out/step_overlap_ab.txt; : > $out
python -m pytest tests/test_update_fluxes.py -m gpu -q 2>&1 | tail -2 | tee -a $out
for ncol in 256 1024 2048 4096 8192 12288 16384; do
  for rep in 1 2; do
    for lim in 0 1; do
      r=$(RRTMGP_HIP_STEP_OVERLAP=$lim python bench.py --ncol $ncol --nlay 72 --aerosols --steps 50 --warmup 5 --no-legs --cpu-sample 0 --fused-step 2>&1 | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M  %.3f ms' % (d['value']/1e6, d['ms_per_step']))")
      echo "ncol $ncol overlap=$lim rep $rep: $r" | tee -a $out
    done
  done
done
r=$(python bench.py --ncol 4096 --nlay 72 --aerosols --steps 50 --warmup 5 --no-legs --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M  %.3f ms' % (d['value']/1e6, d['ms_per_step']))")
echo "ncol 4096 two solve calls: $r" | tee -a $out
