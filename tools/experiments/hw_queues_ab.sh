#!/bin/bash
# Runs on the GPU box: the host-array legs under different numbers of hardware queues per process (GPU_MAX_HW_QUEUES, ROCm
# default 4): streams are multiplexed onto them, and a copy stream that shares a queue with a compute stream waits for its kernels.
for rep in 1 2; do for q in 2 4 8 16; do
  for mode in "--host" "--l2 fused"; do
  GPU_MAX_HW_QUEUES=$q python bench.py $mode --leg x --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q', '$mode', '%.3f M %.2f ms' % (j['value']/1e6, j['ms_per_step']))"
  done
done; done
