#!/bin/bash
# Runs a list of steps on the GPU box, one after the other, each under its own timeout and with its own log
# under gpurun_out/<session>/.  A step that fails goes on to the next one; a step that is KILLED AT ITS TIME
# LIMIT ends the session (no further GPU step after a hang).
#   usage: tools/gpu_session.sh <session> <<'STEPS'
#          name|seconds|command ...
#          STEPS
SESSION=$1
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$SESSION
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
while IFS='|' read -r name secs cmd; do
    [ -z "$name" ] && continue
    echo "== $name (limit ${secs}s): $cmd"
    t0=$(date +%s)
    timeout -k 10 $secs bash -c "$cmd" > $OUT/$name.log 2>&1
    rc=$?
    echo "== $name rc=$rc $(( $(date +%s) - t0 ))s"
    tail -n 6 $OUT/$name.log | cut -c1-600
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "== $name hit its time limit: session ends here"
        exit 1
    fi
done
exit 0
