#!/bin/bash
# Runs on the GPU box: the host-array rate (bench.py --host) for a few chunk sizes of the pipelined host path.
# Usage: tools/experiments/host_ab.sh [chunk columns ...]
for rep in 1 2; do
for ch in "${@:-8192}"; do
  export RRTMGP_HIP_HOST_CHUNK_COLUMNS=$ch
  python bench.py --host --steps 8 --warmup 2 --cpu-sample 0 --no-legs 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('chunk %-8s %.3f M col/s  %.2f ms/step' % ('$ch', j['value']/1e6, j['ms_per_step']))"
done; done
