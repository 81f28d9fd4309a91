#!/bin/bash
# Interleaved A/B of the engine's environment switches on the headline step: bash tools/env_ab.sh <tag> "VAR=val" "VAR2=val" ...
# (each variant and the default twice, alternating; 40 timed steps after 20 warm-up steps)  -> gpurun_out/<tag>/summary.txt
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
run() { # name, env assignment
    env $2 python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-secondary --no-trainer-loop --no-score-gemm 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', d['ms_per_step'], d['ms_per_step_median'], d['config']['loss_last_step'])" | tee -a $out/summary.txt
}
for rep in 1 2; do
    run default "CPC_NOOP=1"
    for v in "$@"; do run "$v" "$v"; done
done
