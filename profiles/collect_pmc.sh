#!/bin/bash
# Profiles bench.py on the GPU box: one rocprofv3 kernel trace with statistics, then one run per PMC pass
# (counters only, never combined with a trace).  usage, from the repository root:
#   profiles/collect_pmc.sh <tag> [wide|ordered|reference] bench.py in that traversal mode (default ordered); BENCH_ARGS in
#                                                          the environment adds arguments (e.g. "--spp-per-step 32")
#   profiles/collect_pmc.sh <tag> script.py args...        that python script (path relative to the repository
#                                                          root) instead, with the same arguments in every pass
# Output: gpurun_out/<tag>/{trace,sq,mem,fetch,write}/ and gpurun_out/<tag>/pmc_summary.{json,txt}
set -u
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TRAVERSAL=""
if [ $# -ge 2 ] && [ "$2" != ordered ] && [ "$2" != reference ] && [ "$2" != wide ]; then
  shift
  PROG=$GRAFT_REPO_ROOT/$1; shift
  TRACE_ARGS=("$@"); PASS_ARGS=("$@")
else
  TRAVERSAL=${2:-ordered}
  PROG=$GRAFT_REPO_ROOT/bench.py
  TRACE_ARGS=(--steps 4 --warmup 1 --cpu-baseline-spp 0 --traversal $TRAVERSAL ${BENCH_ARGS:-}); PASS_ARGS=(--steps 1 --warmup 0 --cpu-baseline-spp 0 --traversal $TRAVERSAL ${BENCH_ARGS:-})
fi
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $PROG "${TRACE_ARGS[@]}" > $OUT/trace.log 2>&1 || exit 1
run_pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $PROG "${PASS_ARGS[@]}" > $OUT/$name.log 2>&1 || return 1
}
run_pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD && \
run_pass mem TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_MISS_sum GRBM_GUI_ACTIVE && \
run_pass fetch FETCH_SIZE && \
run_pass write WRITE_SIZE && \
run_pass ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS && \
python3 $GRAFT_REPO_ROOT/profiles/summarize_pmc.py $OUT $TRAVERSAL > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
