#!/bin/bash
# Runs a list of GPU steps one after another on the GPU box, each under its own timeout, logs under gpurun_out/<tag>/.
# A step that fails with an ordinary error does not stop the list; a step that is KILLED (timeout) does: no further GPU step is
# started after a kill.   usage: tools/gpu_steps.sh <tag> "<seconds>|<name>|<command>" ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
for spec in "$@"; do
    secs=${spec%%|*}; rest=${spec#*|}; name=${rest%%|*}; cmd=${rest#*|}
    echo "=== $name (limit ${secs}s): $cmd" | tee -a "$out/steps.log"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
    rc=$?
    echo "=== $name rc=$rc $(( $(date +%s) - start ))s" | tee -a "$out/steps.log"
    tail -n 6 "$out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "step $name was killed at its limit: stopping here" | tee -a "$out/steps.log"
        exit 1
    fi
done
exit 0
