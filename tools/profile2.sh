#!/bin/bash
# Runs on the GPU box (via gpurun): latency / memory-pipe PMC passes of the default bench workload for one
# library variant.  Run `git rev-parse HEAD > .git_sha` first (the GPU box has no .git).  Usage: tools/profile2.sh <tag> [variant|base]   -> gpurun_out/prof_<tag>/summary.{txt,json}
# (the rocpd databases are deleted after they have been summarised: gpurun copies back at most 64 MiB)
set -u
TAG=${1:-x}; VAR=${2:-base}; 
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
if [ "$VAR" != base ]; then export RRTMGP_HIP_LIBRARY=$REPO/rrtmgp.jl_amd/variants/$VAR.so; fi
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-legs ${BENCH_ARGS:-}"   # BENCH_ARGS: another workload, e.g. "--lw-solver noscat --no-clouds"
SEL="--kernel-include-regex (solve|noscat)_kernel"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
PASSES=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
        "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
        "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"
        "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
        "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT")
# PROFILE_LITE=1: kernel trace + the two HBM passes only (the variant legs of bench.py: profiles/latest_<leg>.json)
if [ -n "${PROFILE_LITE:-}" ]; then PASSES=("FETCH_SIZE" "WRITE_SIZE"); fi
for pass in "${PASSES[@]}"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $pass --kernel-trace $SEL -d $OUT/pmc_$name -o bench -- $BENCH > $OUT/pmc_$name.log 2>&1 || echo "pass failed: $pass" >> $OUT/failed.txt
done
python $REPO/tools/rocprof_summary.py $OUT $OUT/summary.json "$(cat $REPO/.git_sha 2>/dev/null)" > $OUT/summary.txt 2>&1
cp $REPO/.git_sha $OUT/sha.txt 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_*/ $OUT/*.log
tail -3 $OUT/summary.txt
