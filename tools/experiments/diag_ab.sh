#!/bin/bash
# Runs on the GPU box: the one-pass clear-sky-diagnostic instances of the working tree against another library build
# (variants/<name>.so, e.g. from build_rev.sh), same session: cld_frac 1 and 0.5, then the default workload.
OUT=gpurun_out/diag_ab.txt; mkdir -p gpurun_out; : > $OUT
python -m pytest tests/test_clear_sky_diag.py tests/test_update_fluxes.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4 >> $OUT
run() {  # label, library, bench args
  if [ "$2" = base ]; then unset RRTMGP_HIP_LIBRARY; else export RRTMGP_HIP_LIBRARY=$PWD/rrtmgp.jl_amd/variants/$2.so; fi
  python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-legs $3 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-34s %.3f M col/s   LW %.2f ms  SW %.2f ms' % ('$1', j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))" >> $OUT 2>&1
}
for rep in 1 2; do
  for v in base "$@"; do
    run "$v diag cld_frac=1" $v "--clear-sky-diag one-pass"
    run "$v diag cld_frac=0.5" $v "--clear-sky-diag one-pass --cld-frac 0.5"
    run "$v diag aerosols" $v "--clear-sky-diag one-pass --aerosols"
    run "$v default" $v ""
  done
done
cat $OUT
