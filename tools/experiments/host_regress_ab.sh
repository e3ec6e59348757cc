#!/bin/bash
# Runs on the GPU box: the two-solve host leg (bench.py --host) for the library variants given as arguments, same session.
for rep in 1 2; do for v in "$@"; do
  if [ $v = base ]; then unset RRTMGP_HIP_LIBRARY; else export RRTMGP_HIP_LIBRARY=$PWD/rrtmgp.jl_amd/variants/$v.so; fi
  python bench.py --host --leg x --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '--host', '%.3f M %.2f ms' % (j['value']/1e6, j['ms_per_step']))"
done; done
