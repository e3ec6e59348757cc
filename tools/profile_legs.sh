#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + HBM counters of the bench legs whose kernels are not the headline pair
# (tools/profile2.sh with the leg's flags) -> gpurun_out/prof_<round>_<leg>/summary.{txt,json}; copy them to
# profiles/<round>_<leg>_kernels.txt and profiles/latest_<leg>.json (bench.py folds those into the leg's `profiled` block).
# Usage: tools/profile_legs.sh r06
R=${1:-rXX}
export PROFILE_LITE=1
BENCH_ARGS="--clear-sky-diag one-pass" bash tools/profile2.sh ${R}_clear_sky_diag > /dev/null 2>&1
BENCH_ARGS="--clear-sky-diag one-pass --aerosols" bash tools/profile2.sh ${R}_clear_sky_diag_aerosols > /dev/null 2>&1
BENCH_ARGS="--dtype f64" bash tools/profile2.sh ${R}_f64 > /dev/null 2>&1
BENCH_ARGS="--lw-solver noscat --angles 1 --no-clouds --dtype f64 --nlay 60" bash tools/profile2.sh ${R}_noscat_clear_f64 > /dev/null 2>&1
BENCH_ARGS="--aerosols" bash tools/profile2.sh ${R}_aerosols > /dev/null 2>&1
for leg in clear_sky_diag clear_sky_diag_aerosols f64 noscat_clear_f64 aerosols; do echo "== $leg"; sed -n 4,6p gpurun_out/prof_${R}_$leg/summary.txt | cut -c1-150; done
