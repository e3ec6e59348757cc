#!/bin/bash
# Runs on the GPU box (via gpurun): latency / memory-pipe PMC passes of the default bench workload for one
# library variant.  Usage: tools/profile2.sh <tag> [variant|base]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-x}; VAR=${2:-base}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
if [ "$VAR" != base ]; then export RRTMGP_HIP_LIBRARY=$REPO/rrtmgp.jl_amd/variants/$VAR.so; fi
cd /tmp && export TMPDIR=/tmp
[ -f $REPO/gpurun_out/counters_list.txt ] || rocprofv3 -L > $REPO/gpurun_out/counters_list.txt 2>&1
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" \
            "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES_EQ_64" \
            "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
            "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$name -o bench -- $BENCH > $OUT/pmc_$name.log 2>&1 || echo "pass failed: $pass" >> $OUT/failed.txt
done
find $OUT -name "*.db" -size +20M -delete 2>/dev/null
python $REPO/tools/rocprof_summary.py $OUT $OUT/summary.json > $OUT/summary.txt 2>&1
tail -5 $OUT/summary.txt
