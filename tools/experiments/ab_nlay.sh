for rep in 1 2; do for nl in 64 72 96; do for v in base prev; do
  if [ $v = base ]; then unset RRTMGP_HIP_LIBRARY; else export RRTMGP_HIP_LIBRARY=$PWD/rrtmgp.jl_amd/variants/$v.so; fi
  python bench.py --nlay $nl --steps 8 --warmup 2 --cpu-sample 0 --no-legs 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('nlay $nl %-5s %.3f M col/s   LW %.2f ms  SW %.2f ms' % ('$v', j['value'] / 1e6, j['kernels']['lw_solve_kernel_ms'], j['kernels']['sw_solve_kernel_ms']))"
done; done; done
