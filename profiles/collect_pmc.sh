#!/bin/bash
# Collects rocprofv3 PMC passes (one run per pass, counters only) for a short bench.py run.
# usage: profiles/collect_pmc.sh <tag>   (run on the GPU box from the repository root)
set -u
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --spp-per-step 8 --cpu-baseline-spp 0"
run_pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1 || return 1
}
run_pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD && \
run_pass mem TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_sum TCC_REQ_sum TCC_MISS_sum && \
run_pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
python3 $GRAFT_REPO_ROOT/profiles/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
