#!/bin/bash
# Runs on the GPU box: the default workload at several layer counts with the persistent grid capped at 2 / 3 / 4 workgroups per CU.
for nl in "${@:-64}"; do for cap in 4 3 2; do
  export RRTMGP_HIP_MAX_WG_PER_CU=$cap
  python bench.py --nlay $nl --steps 8 --warmup 2 --cpu-sample 0 --no-legs 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('nlay $nl  %d workgroups/CU  %.3f M col/s   LW %.2f ms  SW %.2f ms' % ($cap, j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))"
done; done
