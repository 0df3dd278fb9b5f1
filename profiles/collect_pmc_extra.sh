#!/bin/bash
# Extra PMC passes for the memory pipeline of the traversal kernels (L1 busy time, L1->L2 round-trip latency, address
# unit busy time, instruction mix).  usage, from the repository root: profiles/collect_pmc_extra.sh <tag>
set -u
TAG=${1:-pmcx}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH=$GRAFT_REPO_ROOT/bench.py
run_pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $BENCH --steps 1 --warmup 0 --cpu-baseline-spp 0 > $OUT/$name.log 2>&1 || return 1
}
run_pass l1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE && \
run_pass l1b TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum && \
run_pass mix SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU && \
python3 $GRAFT_REPO_ROOT/profiles/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
grep -A 30 "^ExtendKernel\|^ConnectKernel" $OUT/pmc_summary.txt
