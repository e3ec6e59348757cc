#!/bin/bash
# Runs on the GPU box (round 5): prepare_column with the layer records shared by the four wavefronts (one part of LayerRec per
# wavefront) and the cloud-table reads of prepare_chunk issued at the top of a task (working tree = base) against the previous
# commit's library (variants/prevcol.so): default workload, with aerosols, Float64, config 4.
OUT=gpurun_out/ab_prepcol.txt; mkdir -p gpurun_out; : > $OUT
for args in "" "--aerosols" "--dtype f64" "--clear-sky-diag one-pass" "--ncol 4096 --nlay 72 --aerosols --steps 50"; do
  echo "== bench.py $args" >> $OUT
  BENCH_ARGS="$args" tools/experiments/ab.sh prepcol_tmp base prevcol > /dev/null 2>&1
  cat gpurun_out/ab_prepcol_tmp.txt >> $OUT
done
cat $OUT
