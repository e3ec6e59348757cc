#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the default bench workload.
# Usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$name -o bench -- $BENCH > $OUT/pmc_$name.log 2>&1
done
# keep only the small summaries
find $OUT -name "*.db" -size +20M -delete 2>/dev/null
ls -R $OUT | head -80
