#!/bin/bash
# Runs on the GPU box: two g-points per lane (RRTMGP_HIP_GPL2=1) against one (=0), same session; optional library variants
# (built by build_variants.sh) as further arms: gpl2_ab.sh [variant ...].  Float32 parity tests with GPL2 first.
OUT=gpurun_out/gpl2_ab.txt; mkdir -p gpurun_out; : > $OUT
RRTMGP_HIP_GPL2=1 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -4 >> $OUT
run() {  # label, gpl2 mode, library
  if [ "$3" = base ]; then unset RRTMGP_HIP_LIBRARY; else export RRTMGP_HIP_LIBRARY=$PWD/rrtmgp.jl_amd/variants/$3.so; fi
  RRTMGP_HIP_GPL2=$2 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-legs ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-22s %.3f M col/s   LW %.2f ms  SW %.2f ms' % ('$1', j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))" >> $OUT 2>&1
}
for rep in 1 2; do
  run gpl1 0 base
  run gpl2 1 base
  for v in "$@"; do run gpl2_$v 1 $v; done
done
cat $OUT
