#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the default bench workload.
#   usage: tools/profile_bench.sh <tag> <pass> [bench args...]
#   pass = trace | fetch | write | sq1 | sq2 | sq3 | grbm   (one rocprofv3 run each: counters never share a run with a trace)
set -o pipefail
TAG=$1; PASS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-other-mode --no-also --no-sweep --no-graph-child --no-fp64-check --repeats 1"
case $PASS in
  trace) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 10 --warmup 2 "$@" > $OUT/trace.log 2>&1 ;;
  fetch) rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 3 --warmup 1 "$@" > $OUT/pmc_fetch.log 2>&1 ;;
  write) rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 3 --warmup 1 "$@" > $OUT/pmc_write.log 2>&1 ;;
  sq1) rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $B --steps 3 --warmup 1 "$@" > $OUT/pmc_sq1.log 2>&1 ;;
  sq2) rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -- $B --steps 3 --warmup 1 "$@" > $OUT/pmc_sq2.log 2>&1 ;;
  sq3) rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS --output-format csv -d $OUT/pmc_sq3 -- $B --steps 3 --warmup 1 "$@" > $OUT/pmc_sq3.log 2>&1 ;;
  grbm) rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $B --steps 3 --warmup 1 "$@" > $OUT/pmc_grbm.log 2>&1 ;;
  *) echo "unknown pass $PASS"; exit 2 ;;
esac
rc=$?
tail -n 3 $OUT/*_$PASS.log $OUT/$PASS.log 2>/dev/null | cut -c1-300
exit $rc
