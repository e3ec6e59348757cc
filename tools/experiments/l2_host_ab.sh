#!/bin/bash
# Runs on the GPU box: the host-array Layer-2 step (bench.py --l2 fused) and the two-solve host leg under different chunk
# schedules of the column pipeline, same session.  Output: gpurun_out/l2_host_ab.txt
OUT=gpurun_out/l2_host_ab.txt; mkdir -p gpurun_out; : > $OUT
python -m pytest tests/test_update_fluxes.py tests/test_gpu_parity.py tests/test_abi_contracts.py -m gpu -q -k "pipeline or host or alloc or thread" 2>&1 | tail -3 >> $OUT
run() {  # label, env assignments
  for mode in "--l2 fused" "--host"; do
    env $2 python bench.py $mode --leg x --steps 10 --warmup 3 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-36s %-12s %.3f M  %.2f ms' % ('$1', '$mode', j['value']/1e6, j['ms_per_step']))" >> $OUT
  done
}
for r in 1 2; do
  run "ramp 4k..32k (default)" "X=1"
  run "equal 8192 (round 3)" "RRTMGP_HIP_HOST_CHUNK_COLUMNS=8192"
  run "ramp, system HIP runtime (no torch)" "RRTMGP_HIP_NO_TORCH=1"
done
cat $OUT
