python -m pytest tests/test_strided_views.py tests/test_update_fluxes.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -12
for r in 1 2; do
for v in ramp noramp; do
  if [ $v = noramp ]; then export RRTMGP_HIP_HOST_NO_RAMP=1; else unset RRTMGP_HIP_HOST_NO_RAMP; fi
  python bench.py --l2 fused --leg x --steps 10 --warmup 3 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v fused %.3f M  %.2f ms min %.2f' % (j['value']/1e6, j['ms_per_step'], j['min_ms']))"
  python bench.py --host --leg x --steps 10 --warmup 3 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v host  %.3f M  %.2f ms' % (j['value']/1e6, j['ms_per_step']))"
done; done
for c in 4096 6144 12288 16384; do
  RRTMGP_HIP_HOST_CHUNK_COLUMNS=$c python bench.py --l2 fused --leg x --steps 10 --warmup 3 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $c fused %.3f M  %.2f ms' % (j['value']/1e6, j['ms_per_step']))"
done
