#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + the two HBM PMC passes of the default bench.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-mode --no-also --no-graph-child "$@" > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-also --no-graph-child "$@" > $OUT/pmc_fetch.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-also --no-graph-child "$@" > $OUT/pmc_write.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-also --no-graph-child "$@" > $OUT/pmc_sq.log 2>&1 || exit 4
cd $R
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
